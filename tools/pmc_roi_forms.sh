#!/bin/bash
# HBM-side traffic of the RoI kernel in every form of tools/roi_forms.py: two rocprofv3 --pmc passes per form (FETCH_SIZE,
# WRITE_SIZE: they do not fit one pass), every pass with its own calibration dispatches (k_calib_read<16> / <8>: 512 MiB read
# once at 16 / 8 bytes per lane -> the FETCH_SIZE factor of THIS pass for both lane widths).
#   tools/pmc_roi_forms.sh <outdir>  ->  <outdir>/roi_forms_pmc.json   (copy to profiles/<round>_roi_forms_pmc.json)
set -u
cd "$(dirname "$0")/.."
out=$1
mkdir -p $out
export TMPDIR=/tmp
for i in ${FORMS:-0 1 2 3 4 5 6}; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 180 rocprofv3 --pmc $c -d $out/f${i}_$c --output-format csv -- python3 tools/roi_forms.py --form $i --reps 8 --cold-only > $out/f${i}_$c.log 2>&1
    echo "form $i $c rc=$?"
  done
done
python3 - "$out" <<'PY'
import csv, glob, json, sys
out = sys.argv[1]
def rows(path):
    return [(r['Kernel_Name'], r['Counter_Name'], float(r['Counter_Value'])) for r in csv.DictReader(open(path))]
res = {'forms': {}, 'how': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over tools/roi_forms.py --form i --reps 8 --cold-only; '
       'counters in KiB; read bytes = FETCH_SIZE x the factor measured in the same pass on k_calib_read<16> (float32 maps) or '
       'k_calib_read<8> (float16 maps): 512 MiB read once; WRITE_SIZE taken as it is; means over the launches after the first two'}
for i in range(7):
    f = glob.glob('%s/f%d_FETCH_SIZE/*/*_counter_collection.csv' % (out, i))
    w = glob.glob('%s/f%d_WRITE_SIZE/*/*_counter_collection.csv' % (out, i))
    if not f or not w:
        continue
    fr, wr = rows(f[0]), rows(w[0])
    cal = {}
    for bpl in (16, 8):
        v = [x[2] for x in fr if 'k_calib_read<%d>' % bpl in x[0] and x[1] == 'FETCH_SIZE']
        if v:
            cal[bpl] = (1 << 29) / (sum(v) / len(v) * 1024.0)
    fv = [x[2] for x in fr if 'k_roi_pool' in x[0] and x[1] == 'FETCH_SIZE'][2:]
    wv = [x[2] for x in wr if 'k_roi_pool' in x[0] and x[1] == 'WRITE_SIZE'][2:]
    if not fv or not wv:
        continue
    f16 = i in (1, 5)
    factor = cal.get(8 if f16 else 16, 2.0)
    rd = sum(fv) / len(fv) * 1024.0 * factor
    wb = sum(wv) / len(wv) * 1024.0
    res['forms'][str(i)] = dict(FETCH_SIZE_KiB=sum(fv) / len(fv), WRITE_SIZE_KiB=sum(wv) / len(wv), fetch_factor_16B_lanes=cal.get(16),
                                fetch_factor_8B_lanes=cal.get(8), fetch_factor_used=factor, read_bytes_per_launch=rd,
                                write_bytes_per_launch=wb, hbm_bytes_per_launch=rd + wb, launches=len(fv))
json.dump(res, open(out + '/roi_forms_pmc.json', 'w'), indent=1)
print(json.dumps(res, indent=1))
PY
find $out -name "*_counter_collection.csv" -delete; find $out -name "*_agent_info.csv" -delete

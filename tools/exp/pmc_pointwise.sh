#!/bin/bash
# counter passes over pointwise layer shapes -> <outdir>/pmc.json: MFMA busy, LDS conflicts, HBM-side bytes per launch
#   tools/exp/pmc_pointwise.sh <outdir>
set -u
cd "$(dirname "$0")/../.."
out=$1; shift
mkdir -p $out
export TMPDIR=/tmp
shapes=("conv4_c1 50 84 1024 256 1 30" "conv3_c1 100 167 512 128 1 30" "fc1 1 30000 12544 1024 1 1" "conv4_b1_sc_s2 100 167 512 1024 2 30" "l2 200 334 256 256 1 30")
for sh in "${shapes[@]}"; do
  set -- $sh; name=$1; shift
  i=0
  for cset in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES" \
              "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" \
              "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout -s KILL 120 rocprofv3 --pmc $cset -d $out/${name}_p$i --output-format csv -- python3 tools/exp/pointwise_one.py "$@" > $out/${name}_p$i.log 2>&1
  done
done
python3 - "$out" <<'PY'
import csv, glob, json, os, sys
res = {}
for d in sorted(glob.glob(os.path.join(sys.argv[1], '*_p*'))):
    if not os.path.isdir(d):
        continue
    name = os.path.basename(d).rsplit('_p', 1)[0]
    for f in glob.glob(os.path.join(d, '*', '*_counter_collection.csv')):
        by = {}
        for r in csv.DictReader(open(f)):
            if 'k_pointwise' in r['Kernel_Name']:
                by.setdefault((r['Counter_Name'], r['Kernel_Name'][:40]), []).append(float(r['Counter_Value']))
        for (k, kn), v in by.items():
            res.setdefault(name, {'kernel': kn})[k] = sum(v[2:]) / max(1, len(v[2:]))
for name, r in res.items():
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in r and 'SQ_BUSY_CU_CYCLES' in r:
        r['mfma_busy_fraction_of_simd_cycles'] = r['SQ_VALU_MFMA_BUSY_CYCLES'] / r['SQ_BUSY_CU_CYCLES'] / 4.0
    if 'FETCH_SIZE' in r and 'WRITE_SIZE' in r:
        r['hbm_side_MB'] = (2.0 * r['FETCH_SIZE'] + r['WRITE_SIZE']) * 1024 / 1e6
json.dump(res, open(os.path.join(sys.argv[1], 'pmc.json'), 'w'), indent=1)
print(json.dumps(res, indent=1))
PY
find $out -name "*.csv" -delete

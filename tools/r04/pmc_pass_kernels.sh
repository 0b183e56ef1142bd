#!/bin/bash
# per-kernel counters of a float16 ResNet-101-FPN pass (BATCH images): MFMA-busy, vector-instruction issue, LDS -- by kernel
# name and grid, over the last passes of tools/e2e_bench.py
set -u
cd "$(dirname "$0")/../.."
o=gpurun_out/r04_pass_pmc; rm -rf $o; mkdir -p $o; export TMPDIR=/tmp
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES"; do
  i=$((i+1))
  timeout -s KILL 400 rocprofv3 --pmc $set -d $o/p$i --output-format csv -- python3 tools/e2e_bench.py --dtype fp16 --batch ${BATCH:-30} --steps 4 --warmup 2 > $o/p$i.log 2>&1
  echo "pass $i rc=$?"
done
python3 - $o <<'PY'
import csv, glob, sys, collections, json
o = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(o + '/p*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        key = (r['Kernel_Name'][:70], r['Grid_Size'])
        acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
rows = []
for key, c in acc.items():
    m = {k: sum(v) / len(v) for k, v in c.items()}
    n = len(c.get('GRBM_GUI_ACTIVE', []))
    gui = m.get('GRBM_GUI_ACTIVE', 0.0) / 8.0            # cycles per XCD
    if gui <= 0:
        continue
    rows.append(dict(kernel=key[0], grid=key[1], calls=n, us=gui / 2.4e3, mfma_busy=m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (1024 * gui),
                     valu_issue=4 * m.get('SQ_INSTS_VALU', 0) / (1024 * gui), lds_busy=m.get('SQ_LDS_IDX_ACTIVE', 0) / (256 * gui),
                     lds_conflict=m.get('SQ_LDS_BANK_CONFLICT', 0) / max(1.0, m.get('SQ_LDS_IDX_ACTIVE', 0)), total_us=gui / 2.4e3 * n))
rows.sort(key=lambda r: -r['total_us'])
json.dump(rows, open(o + '/kernels.json', 'w'), indent=1)
print('%-70s %9s %5s %8s %6s %6s %6s %6s' % ('kernel', 'grid', 'calls', 'us@2.4G', 'mfma', 'valu', 'lds', 'confl'))
for r in rows[:40]:
    print('%-70s %9s %5d %8.1f %6.2f %6.2f %6.2f %6.2f' % (r['kernel'], r['grid'], r['calls'], r['us'], r['mfma_busy'], r['valu_issue'], r['lds_busy'], r['lds_conflict']))
PY
find $o -name "*.csv" -delete

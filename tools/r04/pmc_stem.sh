#!/bin/bash
# counters of k_stem_conv7_pool3 (batch 8), a few per pass
set -u
cd "$(dirname "$0")/../.."
o=gpurun_out/r04_stem_pmc; rm -rf $o; mkdir -p $o; export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAIT_ANY"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $set -d $o/p$i --output-format csv -- python3 tools/r04/stem_probe.py 8 > $o/p$i.log 2>&1
  echo "pass $i rc=$?"
done
python3 - $o <<'PY'
import csv, glob, sys, collections
o = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(o + '/p*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'k_stem_conv7' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(acc):
    v = acc[k]
    print('%-34s %16.0f  (%d dispatches)' % (k, sum(v) / len(v), len(v)))
PY
find $o -name "*.csv" -delete

#!/bin/bash
# counters of the split-precision kernel on one layer (separate --pmc passes; no trace domains beside them)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${2:-x3}_pmc
mkdir -p $OUT
SHAPE=${1:-15,200,334,256,512,3}
FORM=${2:-x3}          # x3 | x2
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM SQ_INSTS_SALU"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d $OUT/p$i -o p --output-format csv -- python3 $R/tools/r05/x3_one.py --form $FORM --shape $SHAPE --reps 3 > $OUT/p$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$OUT/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if '_$FORM' in r['Kernel_Name'] and 'split' not in r['Kernel_Name']:
            agg[r['Kernel_Name'][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print('  %-28s %16.0f  (n=%d)' % (c, sum(v) / len(v), len(v)))
PY

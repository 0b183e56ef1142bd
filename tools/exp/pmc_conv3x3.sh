#!/bin/bash
# counter passes over one odet_conv3x3_f16 shape (default: the RPN head's P2 level at batch 8) -> <outdir>/pmc.json
set -u
cd "$(dirname "$0")/../.."
out=$1; shift
mkdir -p $out
export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $set -d $out/p$i --output-format csv -- python3 tools/exp/conv3x3_one.py "$@" > $out/p$i.log 2>&1
  echo "pass $i rc=$? : $set"
done
python3 - "$out" <<'PY'
import csv, glob, json, os, sys
out = {}
for f in sorted(glob.glob(os.path.join(sys.argv[1], 'p*', '*', '*_counter_collection.csv'))):
    rows = [r for r in csv.DictReader(open(f)) if 'k_conv3x3' in r['Kernel_Name']]
    by = {}
    for r in rows:
        by.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
    for k, v in by.items():
        out[k] = sum(v[2:]) / max(1, len(v[2:]))
json.dump(out, open(os.path.join(sys.argv[1], 'pmc.json'), 'w'), indent=1)
print(json.dumps(out, indent=1))
PY
find $out -name "*agent_info*" -delete

#!/bin/bash
# counters of the float16-map RoI launch (tools/roi_forms.py --form 1, 8-image launches, cold) for the library in ODET_LIB_PATH
#   tools/r04/pmc_roi_f16.sh <outdir>
set -u
cd "$(dirname "$0")/../.."
out=$1; mkdir -p $out; export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAVES GRBM_GUI_ACTIVE" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCP_GATE_EN1_sum TCP_TA_TCP_STATE_READ_sum"; do
  i=$((i+1))
  timeout -s KILL 150 rocprofv3 --pmc $set -d $out/p$i --output-format csv -- python3 tools/roi_forms.py --form 1 --reps 8 --cold-only > $out/p$i.log 2>&1
  echo "pass $i rc=$? : $set"
done
python3 - $out <<'PY'
import csv, glob, json, sys
res = {}
for f in sorted(glob.glob(sys.argv[1] + '/p*/*/*_counter_collection.csv')):
    by = {}
    for r in csv.DictReader(open(f)):
        if 'k_roi_pool' in r['Kernel_Name']:
            by.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
    for k, v in by.items():
        v = v[2:] if len(v) > 3 else v
        res[k] = sum(v) / len(v)
json.dump(res, open(sys.argv[1] + '/pmc.json', 'w'), indent=1)
print(json.dumps(res, indent=1))
PY
find $out -name "*_counter_collection.csv" -delete; find $out -name "*_agent_info.csv" -delete

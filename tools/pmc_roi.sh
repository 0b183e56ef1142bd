#!/bin/bash
# Hardware-counter passes over tools/roi_bench.py (8-image cold / warm, one-image cold / warm launches of the RoI kernel).
#   tools/pmc_roi.sh <outdir> [extra roi_bench args]   -> <outdir>/p<i>/..._counter_collection.csv + <outdir>/pmc.json
# Few counters per pass and a hard timeout: an over-subscribed set makes rocprofv3 abort and then hang.
set -u
cd "$(dirname "$0")/.."
out=$1; shift
mkdir -p $out
export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES" \
           "TCC_HIT_sum TCC_MISS_sum" \
           "TCC_REQ_sum TCC_BUSY_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" \
           "WRITE_SIZE"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $set -d $out/p$i --output-format csv -- python3 tools/roi_bench.py --reps 8 "$@" > $out/p$i.log 2>&1
  echo "pass $i rc=$? : $set"
done
python3 tools/pmc_roi_parse.py $out --reps 8 > $out/pmc.json
cat $out/pmc.json

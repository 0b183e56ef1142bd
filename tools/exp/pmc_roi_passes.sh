# memory-side / issue-side PMC passes over tools/roi_ablate.py (30 warm + 30 cold launches of the RoI kernel).
# Two counters per pass and a hard timeout: an over-subscribed counter set makes rocprofv3 abort and then hang.
cd /root/repo
i=0
for set in "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" \
           "TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "TCC_HIT_sum TCC_MISS_sum" \
           "TCC_REQ_sum TCC_BUSY_sum" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL" \
           "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
           "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
           "TA_FLAT_READ_WAVEFRONTS_sum TD_TD_BUSY_sum"; do
  i=$((i+1))
  timeout -s KILL 90 rocprofv3 --pmc $set -d gpurun_out/pmc_roi/p$i --output-format csv -- python3 tools/roi_ablate.py > gpurun_out/pmc_roi/p$i.log 2>&1
  echo "pass $i rc=$? : $set"
done

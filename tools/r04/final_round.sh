#!/bin/bash
# everything profiles/r04_* holds that depends on the final kernels, in ONE GPU call (one box):
#   tools/profile_round.sh r04 (bench line of the driver's command, its rocprofv3 kernel stats, the latency arrangement, the
#   RoI launch's counter traffic), the pass traces at batch 1 / 4 / 30, MFMA-busy at batch 30, the RoI forms + their counters
set -u
cd "$(dirname "$0")/../.."
tools/profile_round.sh r04 > gpurun_out/r04_round.log 2>&1
o=gpurun_out/r04x; rm -rf $o; mkdir -p $o
for b in 1 4 30; do
  BATCH=$b tools/exp/e2e_pass_trace.sh > /dev/null 2>&1; cp gpurun_out/e2e_pass/pass.txt $o/e2e_fp16_b${b}_pass_trace.txt
done
BATCH=30 tools/exp/mfma_busy_run.sh > /dev/null 2>&1; cp gpurun_out/mfma_busy/busy.json $o/e2e_mfma_busy_fp16_b30.json
tools/pmc_roi_forms.sh $o/pmc_forms > $o/pmc_forms.log 2>&1; cp $o/pmc_forms/roi_forms_pmc.json $o/roi_forms_pmc.json
timeout 600 python3 tools/roi_forms.py --pmc $o/roi_forms_pmc.json > $o/roi_forms.json 2> $o/roi_forms.err
find $o/pmc_forms -name "*.csv" -delete
ls -la $o gpurun_out/r04 | head -40
tail -c 1200 gpurun_out/r04/bench.json

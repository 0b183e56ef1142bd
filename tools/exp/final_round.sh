#!/bin/bash
# everything profiles/r03_* holds, in one GPU call
cd "$(dirname "$0")/../.."
tools/profile_round.sh r03 > gpurun_out/r03_round.log 2>&1
mkdir -p gpurun_out/r03x
BATCH=30 tools/exp/e2e_pass_trace.sh > /dev/null 2>&1; cp gpurun_out/e2e_pass/pass.txt gpurun_out/r03x/e2e_fp16_b30_pass_trace.txt
BATCH=8 tools/exp/e2e_pass_trace.sh > /dev/null 2>&1; cp gpurun_out/e2e_pass/pass.txt gpurun_out/r03x/e2e_fp16_b8_pass_trace.txt
DT=fp32 BATCH=15 tools/exp/e2e_pass_trace.sh > /dev/null 2>&1; cp gpurun_out/e2e_pass/pass.txt gpurun_out/r03x/e2e_fp32_b15_pass_trace.txt
BATCH=30 EXTRA="--model c4 --depth 50" tools/exp/e2e_pass_trace.sh > /dev/null 2>&1; cp gpurun_out/e2e_pass/pass.txt gpurun_out/r03x/e2e_c4_fp16_b30_pass_trace.txt
BATCH=32 EXTRA="--model vgg16 --h 600 --w 800" tools/exp/e2e_pass_trace.sh > /dev/null 2>&1; cp gpurun_out/e2e_pass/pass.txt gpurun_out/r03x/e2e_vgg16_fp16_b32_pass_trace.txt
BATCH=30 tools/exp/mfma_busy_run.sh > /dev/null 2>&1; cp gpurun_out/mfma_busy/busy.json gpurun_out/r03x/e2e_mfma_busy_fp16_b30.json
python3 bench.py --maps f16 --no-cpu-baseline --no-e2e --steps 20 --warmup 5 > gpurun_out/r03x/bench_f16maps.json 2> gpurun_out/r03x/bench_f16maps.err
tools/exp/bench_timeline.sh distinct > /dev/null 2>&1; tools/exp/bench_timeline.sh clustered > /dev/null 2>&1
cp gpurun_out/timeline_distinct/timeline.txt gpurun_out/r03x/bench_timeline_distinct.txt; cp gpurun_out/timeline_clustered/timeline.txt gpurun_out/r03x/bench_timeline_clustered.txt
ls -la gpurun_out/r03x gpurun_out/r03 | head -40

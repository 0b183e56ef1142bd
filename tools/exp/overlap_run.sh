#!/bin/bash
# kernel trace of a short pipelined bench run -> tools/exp/overlap_report.py (run on the GPU box)
set -u
cd "$(dirname "$0")/../.."
out=gpurun_out/overlap; rm -rf $out; mkdir -p $out; export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace -d $out/tr --output-format csv -- python3 bench.py --gpus 1 --steps ${STEPS:-12} --warmup 2 --no-e2e --no-cpu-baseline "$@" > $out/run.log 2>&1
f=$(find $out -name "*_kernel_trace.csv" | head -1)
python3 tools/exp/overlap_report.py "$f" > $out/report.txt 2>&1
find $out -name "*_kernel_trace.csv" -delete; find $out -name "*_agent_info.csv" -delete
cat $out/report.txt

#!/bin/bash
# kernel trace of the bench's hot-path loop for one score distribution -> timeline statistics (the trace itself is deleted)
cd "$(dirname "$0")/../.."
sc=${1:-clustered}; out=gpurun_out/timeline_$sc
# (clustered scores: start at the plan the bench's re-plan ladder settles on, so that no re-plan falls into the traced window)
[ "$sc" = clustered ] && EXTRA="--nms-first-chunk 2560 ${EXTRA:-}"; rm -rf $out; mkdir -p $out; export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace -d $out/tr --output-format csv -- python3 bench.py --scores $sc --no-second-distribution --no-cpu-baseline --no-e2e --steps 6 --warmup 2 ${EXTRA:-} > $out/run.log 2>&1
f=$(find $out -name "*_kernel_trace.csv" | head -1)
python3 tools/exp/bench_timeline.py "$f" | tee $out/timeline.txt
rm -rf $out/tr

#!/bin/bash
cd "$(dirname "$0")/../.."
out=gpurun_out/clu; rm -rf $out; mkdir -p $out; export TMPDIR=/tmp
for sc in distinct clustered; do
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $out/$sc --output-format csv -- python3 bench.py --scores $sc --no-second-distribution --no-cpu-baseline --no-e2e --steps 10 --warmup 3 > $out/$sc.log 2>&1
tail -1 $out/$sc.log | cut -c1-200
find $out/$sc -name "*_kernel_trace.csv" -delete; find $out/$sc -name "*_agent_info.csv" -delete
done

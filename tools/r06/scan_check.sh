#!/bin/bash
# round 6: the parallel NMS iteration of k_nms_scan<true> -- parity (every NMS / proposal / full-size test) and the latency arrangement
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "nms or proposal or full_size or fpn or hot or batched" 2>&1 | tail -5
A="--streams 1 --batch 1 --rounds-per-step 768 --no-cpu-baseline --no-e2e --no-config5"
python3 bench.py $A --steps 10 --warmup 2 > gpurun_out/r06_s1b1_fp.json 2>/dev/null
python3 -c "
import json; d=json.load(open('gpurun_out/r06_s1b1_fp.json')); print('latency us/img distinct', 1e6/d['value'], 'clustered', 1e6/d['value_clustered'])"
rm -rf gpurun_out/r06_s1b1_stats
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r06_s1b1_stats --output-format csv -- python3 bench.py $A --steps 2 --warmup 1 > /dev/null 2>&1
f=$(find gpurun_out/r06_s1b1_stats -name "*_kernel_stats.csv" | head -1); head -12 $f | cut -c1-110
find gpurun_out/r06_s1b1_stats -name "*_kernel_trace.csv" -delete

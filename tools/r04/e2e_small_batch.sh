#!/bin/bash
# round 4: end-to-end rates + pass traces at the BASELINE configs' own batch sizes (1, 4; float16, ResNet-101-FPN 800x1333)
set -u
cd "$(dirname "$0")/../.."
o=gpurun_out/r04_e2e; mkdir -p $o
for b in 1 2 4 8; do
  timeout 600 python3 tools/e2e_bench.py --batch $b --graph > $o/e2e_b${b}_graph.json 2> $o/e2e_b${b}_graph.err
  timeout 600 python3 tools/e2e_bench.py --batch $b > $o/e2e_b${b}.json 2> $o/e2e_b${b}.err
done
BATCH=1 bash tools/exp/e2e_pass_trace.sh > /dev/null 2>&1; cp gpurun_out/e2e_pass/pass.txt $o/pass_b1.txt
BATCH=4 bash tools/exp/e2e_pass_trace.sh > /dev/null 2>&1; cp gpurun_out/e2e_pass/pass.txt $o/pass_b4.txt
tail -n 2 $o/*.json

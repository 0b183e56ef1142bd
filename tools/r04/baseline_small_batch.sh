#!/bin/bash
# round 4 baseline: where batch 1 / 4 stand before the small-M kernels (library vs own per layer, e2e rates, pass traces)
set -u
cd "$(dirname "$0")/../.."
o=gpurun_out/r04_base; mkdir -p $o
export MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD=0
for b in 1 2 4; do
  timeout 300 python3 tools/exp/conv3x3_layers.py $b > $o/c3_layers_b$b.json 2> $o/c3_layers_b$b.err
  timeout 300 python3 tools/exp/block_tail_layers.py $b > $o/tail_layers_b$b.txt 2>&1
done
timeout 300 python3 tools/exp/pointwise_layers.py 1 > $o/pw_layers_b1.txt 2>&1
timeout 600 python3 tools/e2e_bench.py --batch 1 --graph --miopen-find > $o/e2e_b1_graph.json 2> $o/e2e_b1_graph.err
timeout 600 python3 tools/e2e_bench.py --batch 1 --miopen-find > $o/e2e_b1_eager.json 2> $o/e2e_b1_eager.err
timeout 600 python3 tools/e2e_bench.py --batch 4 --miopen-find > $o/e2e_b4.json 2> $o/e2e_b4.err
timeout 600 python3 tools/e2e_bench.py --batch 4 --graph --miopen-find > $o/e2e_b4_graph.json 2> $o/e2e_b4_graph.err
BATCH=1 bash tools/exp/e2e_pass_trace.sh > /dev/null 2>&1; cp gpurun_out/e2e_pass/pass.txt $o/pass_b1.txt
BATCH=4 bash tools/exp/e2e_pass_trace.sh > /dev/null 2>&1; cp gpurun_out/e2e_pass/pass.txt $o/pass_b4.txt
tail -n 3 $o/*.json $o/tail_layers_b1.txt

#!/bin/bash
# round 4: the fused RpnHead as ONE launch (a workgroup walks the channel tiles of its slab): parity, fuzz, end-to-end rates
set -u
cd "$(dirname "$0")/../.."
o=gpurun_out/r04_rpn_fused; mkdir -p $o
timeout 900 python3 -m pytest tests/test_detector.py -x -q -m gpu -k "rpn_head or no_library or rpn" > $o/pytest.txt 2>&1; tail -n 3 $o/pytest.txt
timeout 900 python3 tools/fuzz_conv.py --cases 630 --seed 41 > $o/fuzz.txt 2>&1; tail -n 2 $o/fuzz.txt
timeout 300 python3 tools/exp/rpn_fused_quick.py > $o/quick.txt 2>&1; tail -n 6 $o/quick.txt
for b in 1 4; do
  timeout 600 python3 tools/e2e_bench.py --batch $b --graph > $o/e2e_b${b}_graph.json 2> $o/e2e_b${b}_graph.err; tail -n 1 $o/e2e_b${b}_graph.json
done
timeout 600 python3 tools/e2e_bench.py --batch 30 --steps 8 --warmup 3 > $o/e2e_b30.json 2> $o/e2e_b30.err; tail -n 1 $o/e2e_b30.json

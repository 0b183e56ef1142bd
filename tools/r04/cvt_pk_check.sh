#!/bin/bash
# round 4: v_cvt_pk_f16_f32 in the convolution epilogues: exactness (tests + fuzz), end-to-end rates, pass trace at 30 images
set -u
cd "$(dirname "$0")/../.."
o=gpurun_out/r04_cvtpk; mkdir -p $o
timeout 1500 python3 -m pytest tests/test_detector.py -x -q -m gpu > $o/pytest.txt 2>&1; tail -n 2 $o/pytest.txt
timeout 900 python3 tools/fuzz_conv.py --cases 900 --seed 555 > $o/fuzz.txt 2>&1; tail -n 1 $o/fuzz.txt
for b in 1 8; do timeout 600 python3 tools/e2e_bench.py --batch $b --graph 2>/dev/null | tail -n 1 | cut -c1-120; done
timeout 600 python3 tools/e2e_bench.py --batch 30 --steps 10 --warmup 3 2>/dev/null | tail -n 1 | cut -c1-260
timeout 600 python3 tools/e2e_bench.py --batch 30 --steps 10 --warmup 3 --model c4 --depth 50 2>/dev/null | tail -n 1 | cut -c1-120
timeout 600 python3 tools/e2e_bench.py --batch 32 --steps 10 --warmup 3 --model vgg16 --h 600 --w 800 2>/dev/null | tail -n 1 | cut -c1-120
BATCH=30 tools/exp/e2e_pass_trace.sh > /dev/null 2>&1; cp gpurun_out/e2e_pass/pass.txt $o/pass_b30.txt; tail -n 1 $o/pass_b30.txt

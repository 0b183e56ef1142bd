#!/bin/bash
# round 4: the persistent stem kernel: parity (tests + fuzz), its time inside a 30-image pass, end-to-end rates
set -u
cd "$(dirname "$0")/../.."
o=gpurun_out/r04_stem; mkdir -p $o
timeout 900 python3 -m pytest tests/test_detector.py -x -q -m gpu -k "stem or no_library or features" > $o/pytest.txt 2>&1; tail -n 2 $o/pytest.txt
timeout 900 python3 tools/fuzz_conv.py --cases 360 --seed 77 > $o/fuzz.txt 2>&1; tail -n 1 $o/fuzz.txt
BATCH=30 tools/exp/e2e_pass_trace.sh > /dev/null 2>&1; grep -m2 "k_stem" gpurun_out/e2e_pass/pass.txt; tail -n 1 gpurun_out/e2e_pass/pass.txt
BATCH=1 tools/exp/e2e_pass_trace.sh > /dev/null 2>&1; grep -m2 "k_stem" gpurun_out/e2e_pass/pass.txt; tail -n 1 gpurun_out/e2e_pass/pass.txt
BATCH=8 tools/exp/e2e_pass_trace.sh > /dev/null 2>&1; grep -m2 "k_stem" gpurun_out/e2e_pass/pass.txt; tail -n 1 gpurun_out/e2e_pass/pass.txt

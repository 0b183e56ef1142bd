#!/bin/bash
# One pass of everything profiles/ holds for a round (run on the GPU box through gpurun):
#   tools/profile_round.sh <tag>      -> gpurun_out/<tag>/{bench.json, bench_s1b1.json, stats_default/, stats_s1b1/, pmc_fetch/, pmc_write/}
# rocprofv3 always wraps python3 itself; every profiler run has a hard timeout (an over-subscribed counter set
# makes rocprofv3 abort and then hang in its finaliser).
set -u
tag=${1:-prof}
cd "$(dirname "$0")/.."
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
python3 bench.py > $out/bench.json 2> $out/bench.err
python3 bench.py --streams 1 --batch 1 --no-cpu-baseline > $out/bench_s1b1.json 2> $out/bench_s1b1.err
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $out/stats_default --output-format csv -- python3 bench.py --no-cpu-baseline --steps 808 --warmup 96 > $out/stats_default.log 2>&1
# (808 = 8 isolated one-image steps + 100 launches of 8 images, warm-up 96 = 12 launches: every 1000-workgroup RoI
#  dispatch of the trace is one of the isolated steps roofline.kernel_ms is measured on)
python3 tools/roi_launch_shapes.py $out/stats_default/*/*_kernel_trace.csv --out $out/roi_launch_shapes.json > /dev/null
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $out/stats_s1b1 --output-format csv -- python3 bench.py --no-cpu-baseline --streams 1 --batch 1 --steps 400 --warmup 50 > $out/stats_s1b1.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE -d $out/pmc_fetch --output-format csv -- python3 bench.py --no-cpu-baseline --streams 1 --batch 1 --steps 40 --warmup 5 > $out/pmc_fetch.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE -d $out/pmc_write --output-format csv -- python3 bench.py --no-cpu-baseline --streams 1 --batch 1 --steps 40 --warmup 5 > $out/pmc_write.log 2>&1
find $out -name "*.csv" | head -20
cut -c1-300 $out/bench.json

#!/bin/bash
# One pass of everything profiles/ holds for a round (run on the GPU box through gpurun):
#   tools/profile_round.sh <tag>      -> gpurun_out/<tag>/{bench.json, bench_s1b1.json, stats_default/, stats_s1b1/, pmc_fetch/, pmc_write/}
# rocprofv3 always wraps python3 itself; every profiler run has a hard timeout (an over-subscribed counter set
# makes rocprofv3 abort and then hang in its finaliser).
set -u
tag=${1:-prof}
cd "$(dirname "$0")/.."
out=gpurun_out/$tag
rm -rf $out
mkdir -p $out
export TMPDIR=/tmp
python3 bench.py > $out/bench.json 2> $out/bench.err
python3 bench.py --streams 1 --batch 1 --no-cpu-baseline > $out/bench_s1b1.json 2> $out/bench_s1b1.err
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $out/stats_default --output-format csv -- python3 bench.py --no-cpu-baseline --steps 800 --warmup 96 > $out/stats_default.log 2>&1
python3 tools/roi_launch_shapes.py $out/stats_default/*/*_kernel_trace.csv --out $out/roi_launch_shapes.json > /dev/null
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $out/stats_s1b1 --output-format csv -- python3 bench.py --no-cpu-baseline --streams 1 --batch 1 --steps 400 --warmup 50 > $out/stats_s1b1.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE -d $out/pmc_fetch --output-format csv -- python3 bench.py --no-cpu-baseline --streams 1 --batch 8 --steps 160 --warmup 24 > $out/pmc_fetch.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE -d $out/pmc_write --output-format csv -- python3 bench.py --no-cpu-baseline --streams 1 --batch 8 --steps 160 --warmup 24 > $out/pmc_write.log 2>&1
python3 tools/pmc_traffic.py --fetch $out/pmc_fetch/*/*_counter_collection.csv --write $out/pmc_write/*/*_counter_collection.csv --kernel k_roi_pool --workload fpn_hot_path_800x1333_r101fpn_distinct --images-per-launch 8 --out $out/roi_pool_traffic.json > /dev/null
find $out -name "*.csv" | head -20
cut -c1-300 $out/bench.json

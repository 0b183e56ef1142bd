#!/bin/bash
# One pass of everything profiles/ holds for a round (run on the GPU box through gpurun):
#   tools/profile_round.sh <tag>   -> gpurun_out/<tag>/{bench.json, bench_s1b1.json, stats_driver/, stats_s1b1/, pmc_fetch/, pmc_write/, ...}
# rocprofv3 always wraps python3 itself; every profiler run has a hard timeout (an over-subscribed counter set
# makes rocprofv3 abort and then hang in its finaliser).
set -u
tag=${1:-prof}
cd "$(dirname "$0")/.."
out=gpurun_out/$tag
rm -rf $out
mkdir -p $out
export TMPDIR=/tmp
# 1. the driver's command, un-profiled
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err
# 2. the latency arrangement (one image per dispatch sequence, nothing overlaps)
python3 bench.py --streams 1 --batch 1 --rounds-per-step 768 --steps 10 --warmup 2 --no-cpu-baseline --no-e2e > $out/bench_s1b1.json 2> $out/bench_s1b1.err
# 3. kernel stats of EXACTLY the driver's command (its roofline samples run under the kernel name k_roi_pool<.., 1>:
#    that row's average is roofline.kernel_ms)
timeout -s KILL 1800 rocprofv3 --kernel-trace --stats -d $out/stats_driver --output-format csv -- python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/stats_driver.log 2>&1
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $out/stats_s1b1 --output-format csv -- python3 bench.py --streams 1 --batch 1 --rounds-per-step 768 --steps 2 --warmup 1 --no-cpu-baseline --no-e2e > $out/stats_s1b1.log 2>&1
# 4. HBM-side traffic of the 8-image RoI launch (separate --pmc passes; FETCH_SIZE x 2 on gfx950)
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $out/pmc_fetch --output-format csv -- python3 bench.py --no-cpu-baseline --no-e2e --streams 1 --batch 8 --rounds-per-step 4 --steps 4 --warmup 1 > $out/pmc_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $out/pmc_write --output-format csv -- python3 bench.py --no-cpu-baseline --no-e2e --streams 1 --batch 8 --rounds-per-step 4 --steps 4 --warmup 1 > $out/pmc_write.log 2>&1
# (the run has two workloads: rows are told apart by the kernel's template arguments)
python3 tools/pmc_traffic.py --fetch $out/pmc_fetch/*/*_counter_collection.csv --write $out/pmc_write/*/*_counter_collection.csv --kernel "k_roi_pool<1, 1, float" --workload fpn_hot_path_800x1333_r101fpn_distinct --images-per-launch 8 --out $out/roi_pool_traffic.json > /dev/null
python3 tools/pmc_traffic.py --fetch $out/pmc_fetch/*/*_counter_collection.csv --write $out/pmc_write/*/*_counter_collection.csv --kernel "k_roi_pool<1, 1, __half" --workload fpn_hot_path_1333x1333_r101fpn_81cls_f16maps --images-per-launch 8 --out $out/roi_pool_traffic_config5.json > /dev/null
# the per-dispatch traces are tens of MB: only the summaries travel back
find $out -name "*_kernel_trace.csv" -delete; find $out -name "*_agent_info.csv" -delete
find $out -name "*_kernel_stats.csv" | head
cut -c1-300 $out/bench.json
cut -c1-200 $out/bench_s1b1.json

#!/bin/bash
# round 6: everything profiles/r06_* holds that comes from the final tree, in one gpurun call.
#   tools/r06/profile_round6.sh  ->  gpurun_out/r06/...
# rocprofv3 always wraps python3 itself.  bench.py starts its detail process (config 5, e2e legs, gates) as a child: the
# profiler follows it (one more *_kernel_stats.csv / *_counter_collection.csv per process); the counter passes run without it.
set -u
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
out=gpurun_out/r06
rm -rf $out; mkdir -p $out
# 1. the driver's command, un-profiled: the line and its side file
python3 bench.py --gpus 1 --steps 20 --warmup 5 --detail-out $out/bench_detail_driver_cmd.json > $out/bench_driver_cmd.json 2> $out/bench_driver_cmd.err
# 2. kernel stats of EXACTLY the driver's command (row k_roi_pool<1, 1, float, 1> = roofline.kernel_ms of that run's own line)
timeout -s KILL 900 rocprofv3 --kernel-trace --stats -d $out/stats_driver --output-format csv -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --detail-out $out/bench_detail_profiled.json > $out/bench_profiled.json 2> $out/stats_driver.log
# 3. the latency arrangement (one stream, one image per dispatch sequence)
A="--streams 1 --batch 1 --rounds-per-step 768 --no-cpu-baseline --no-e2e --no-config5"
python3 bench.py $A --steps 10 --warmup 2 > $out/bench_streams1_batch1.json 2> $out/bench_s1b1.err
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $out/stats_s1b1 --output-format csv -- python3 bench.py $A --steps 2 --warmup 1 > $out/stats_s1b1.log 2>&1
# 4. HBM-side traffic of the 8-image RoI launch (separate --pmc passes; FETCH_SIZE x 2 on gfx950), float32 maps and config 5's
P="--no-cpu-baseline --no-e2e --no-config5 --no-second-distribution --streams 1 --batch 8 --rounds-per-step 4 --steps 4 --warmup 1"
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $out/pmc_fetch --output-format csv -- python3 bench.py $P > $out/pmc_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $out/pmc_write --output-format csv -- python3 bench.py $P > $out/pmc_write.log 2>&1
python3 tools/pmc_traffic.py --fetch $out/pmc_fetch/*/*_counter_collection.csv --write $out/pmc_write/*/*_counter_collection.csv --kernel "k_roi_pool<1, 1, float" --workload fpn_hot_path_800x1333_r101fpn_distinct --images-per-launch 8 --out $out/roi_pool_traffic.json > /dev/null
# 4b. the same for config 5's float16 launch: the detail process alone (it IS bench.py's config-5 code), no e2e legs
C5="--detail-child /tmp/r06_c5_detail.json --no-e2e --time-budget 120 --streams 1 --batch 8 --rounds-per-step 4 --steps 8 --warmup 2"
timeout -s KILL 180 rocprofv3 --pmc FETCH_SIZE -d $out/pmc_fetch_c5 --output-format csv -- python3 bench.py $C5 > $out/pmc_fetch_c5.log 2>&1
timeout -s KILL 180 rocprofv3 --pmc WRITE_SIZE -d $out/pmc_write_c5 --output-format csv -- python3 bench.py $C5 > $out/pmc_write_c5.log 2>&1
python3 tools/pmc_traffic.py --fetch $out/pmc_fetch_c5/*/*_counter_collection.csv --write $out/pmc_write_c5/*/*_counter_collection.csv --kernel "k_roi_pool<1, 1, __half" --workload fpn_hot_path_1333x1333_r101fpn_81cls_f16maps --images-per-launch 8 --out $out/roi_pool_traffic_config5.json > /dev/null
# 5. the accuracy gates at full size (4096 / 6144 / 8192 scenes; the default run cuts them to its time budget)
python3 bench.py --gpus 1 --steps 20 --warmup 5 --time-budget 900 --detail-out $out/bench_detail_full_gates.json > $out/bench_full_gates.json 2> $out/bench_full_gates.err
# only the summaries travel back
# (the driver's command: one stats file per process -- the parent's holds the isolated RoI launches k_roi_pool<1, 1, float, 1>)
for f in $(find $out/stats_driver -name "*_kernel_stats.csv" | sort); do
  if grep -q "k_roi_pool<1, 1, float, 1>" "$f"; then cp "$f" $out/kernel_stats_driver_cmd.csv; else cp "$f" $out/kernel_stats_driver_cmd_child.csv; fi
done
for f in $(find $out/stats_s1b1 -name "*_kernel_stats.csv" | sort | head -1); do cp "$f" $out/kernel_stats_streams1_batch1.csv; done
find $out -name "*_kernel_trace.csv" -delete; find $out -name "*_agent_info.csv" -delete; find $out -name "*_counter_collection.csv" -delete
ls -la $out | head -40
cut -c1-400 $out/bench_driver_cmd.json; echo; tail -c 1200 $out/bench_full_gates.json

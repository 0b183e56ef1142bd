#!/bin/bash
# round 4: average k_postops time (latency arrangement, distinct scores) of the early-exit builds of tools/r04/postops_probe.sh
# and of the product library, on one box:   tools/r04/postops_probe_run.sh   (on the GPU box, after postops_probe.sh here)
set -u
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
for v in p1 p2 p3 p4 full; do
  if [ $v != full ]; then export ODET_LIB_PATH=$PWD/tools/exp/libodet_po_$v.so; else unset ODET_LIB_PATH; fi
  rm -rf gpurun_out/po_$v
  timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d gpurun_out/po_$v --output-format csv -- python3 bench.py --streams 1 --batch 1 --rounds-per-step 768 --no-cpu-baseline --no-e2e --no-config5 --no-second-distribution --steps 2 --warmup 1 > /dev/null 2>&1
  f=$(find gpurun_out/po_$v -name "*_kernel_stats.csv" | head -1)
  echo "$v $(grep k_postops "$f" | head -1 | cut -d, -f1-4)"
  find gpurun_out/po_$v -name "*_kernel_trace.csv" -delete; find gpurun_out/po_$v -name "*_agent_info.csv" -delete
done

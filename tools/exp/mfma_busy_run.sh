#!/bin/bash
# MFMA-busy of a float16 ResNet-101-FPN pass at batch ${BATCH:-8} (one --pmc pass over tools/e2e_bench.py) -> gpurun_out/mfma_busy/busy.json
set -u
cd "$(dirname "$0")/../.."
out=gpurun_out/mfma_busy; rm -rf $out; mkdir -p $out; export TMPDIR=/tmp
timeout -s KILL 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE -d $out/pmc --output-format csv -- python3 tools/e2e_bench.py --dtype fp16 --batch ${BATCH:-8} --steps ${STEPS:-30} --warmup 5 > $out/run.log 2>&1
f=$(find $out -name "*_counter_collection.csv" | head -1)
python3 tools/mfma_busy.py "$f" --tail 0.5 --out $out/busy.json > /dev/null
find $out -name "*_counter_collection.csv" -delete; find $out -name "*_agent_info.csv" -delete
cat $out/busy.json | head -60

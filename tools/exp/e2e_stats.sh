#!/bin/bash
# kernel stats of the float32 ResNet-101-FPN detector end to end (batch 4), steady state: the find-mode searches
# run in the warm-up; the summary splits the time by kernel family.  Run on the GPU box.
set -u
cd "$(dirname "$0")/../.."
out=gpurun_out/e2e_${DT:-fp32}_stats
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
timeout -s KILL 900 rocprofv3 --kernel-trace --stats -d $out/stats --output-format csv -- python3 tools/e2e_bench.py --dtype ${DT:-fp32} --batch ${BATCH:-4} --steps ${STEPS:-40} --warmup 5 > $out/run.log 2>&1
find $out -name "*_kernel_trace.csv" -delete; find $out -name "*_agent_info.csv" -delete
tail -3 $out/run.log | cut -c1-600
f=$(find $out -name "*_kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:40]:
    print('%8.1f ms %6d calls %9.1f us  %5.2f%%  %s' % (float(r['TotalDurationNs']) / 1e6, int(r['Calls']), float(r['AverageNs']) / 1e3,
          100 * float(r['TotalDurationNs']) / tot, r['Name'][:110]))
print('total %.1f ms' % (tot / 1e6))
PY

set -u
cd "$(dirname "$0")/../.."
out=gpurun_out/s1b8; rm -rf $out; mkdir -p $out; export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $out/st --output-format csv -- python3 bench.py --streams 1 --batch 8 --rounds-per-step 96 --steps 2 --warmup 1 --no-cpu-baseline --no-e2e > $out/run.log 2>&1
find $out -name "*_kernel_trace.csv" -delete; find $out -name "*_agent_info.csv" -delete
f=$(find $out -name "*_kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print('%8.1f us %6d  %s' % (float(r['AverageNs']) / 1e3, int(r['Calls']), r['Name'][:60]))
PY
tail -1 $out/run.log | cut -c1-120

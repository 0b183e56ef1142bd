#!/bin/bash
# round 4: the latency arrangement (one stream, one image per dispatch sequence): us per image and the kernel stats
#   tools/r04/latency_arrangement.sh <outdir>
set -u
cd "$(dirname "$0")/../.."
out=$1; mkdir -p $out; export TMPDIR=/tmp
A="--streams 1 --batch 1 --rounds-per-step 768 --no-cpu-baseline --no-e2e --no-config5"
python3 bench.py $A --steps 10 --warmup 2 > $out/bench_s1b1.json 2> $out/bench_s1b1.err
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $out/stats_s1b1 --output-format csv -- python3 bench.py $A --steps 2 --warmup 1 > $out/stats_s1b1.log 2>&1
f=$(find $out/stats_s1b1 -name "*_kernel_stats.csv" | head -1); cp "$f" $out/kernel_stats_streams1_batch1.csv
find $out -name "*_kernel_trace.csv" -delete; find $out -name "*_agent_info.csv" -delete
python3 - $out <<'PY'
import csv, json, sys
o = sys.argv[1]
d = json.loads(open(o + '/bench_s1b1.json').read().strip().splitlines()[-1])
print('distinct %.1f img/s = %.1f us/image; clustered %.1f img/s = %.1f us/image' % (d['value'], 1e6 / d['value'], d['value_clustered'], 1e6 / d['value_clustered']))
for r in list(csv.DictReader(open(o + '/kernel_stats_streams1_batch1.csv')))[:14]:
    print('%-64s calls %6s avg %7.1f us' % (r['Name'][:64], r['Calls'], float(r['AverageNs']) / 1e3))
PY

#!/bin/bash
# kernel names + total time per name of a steady-state run of a detector family (kernel trace of tools/e2e_bench.py):
#   MODEL=c4|vgg16 BATCH=30 tools/exp/e2e_pass_trace_model.sh  -> gpurun_out/e2e_pass_$MODEL/summary.txt
set -u
cd "$(dirname "$0")/../.."
m=${MODEL:-c4}
out=gpurun_out/e2e_pass_$m; rm -rf $out; mkdir -p $out; export TMPDIR=/tmp
extra=""; [ "$m" = "vgg16" ] && extra="--h 600 --w 800"; [ "$m" = "c4" ] && extra="--depth 50"
timeout -s KILL 600 rocprofv3 --kernel-trace -d $out/tr --output-format csv -- python3 tools/e2e_bench.py --model $m $extra --dtype ${DT:-fp16} --batch ${BATCH:-30} --steps 8 --warmup 5 > $out/run.log 2>&1
f=$(find $out -name "*_kernel_trace.csv" | head -1)
python3 - "$f" > $out/summary.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = rows[int(len(rows) * 0.6):]                 # steady state: the last 40 % of the dispatches
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    a = agg[r['Kernel_Name'][:90]]; a[0] += 1; a[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
tot = sum(v[1] for v in agg.values())
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    lib = any(t in k for t in ('Cijk', '_ZN2ck', 'igemm', 'miopen', 'MIOpen', 'naive'))
    print('%9.1f us %5.1f%% x%-4d %s%s' % (v[1], 100 * v[1] / tot, v[0], 'LIB ' if lib else '    ', k))
PY
find $out -name "*_kernel_trace.csv" -delete; find $out -name "*_agent_info.csv" -delete
grep LIB $out/summary.txt | head -20; head -12 $out/summary.txt

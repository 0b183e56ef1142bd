#!/bin/bash
# kernel sequence of ONE steady-state pass of the end-to-end detector (kernel trace, last pass): name, duration.
set -u
cd "$(dirname "$0")/../.."
out=gpurun_out/e2e_pass; rm -rf $out; mkdir -p $out; export TMPDIR=/tmp
timeout -s KILL 900 rocprofv3 --kernel-trace -d $out/tr --output-format csv -- python3 tools/e2e_bench.py --dtype ${DT:-fp16} --batch ${BATCH:-8} --steps 6 --warmup 5 ${EXTRA:-} > $out/run.log 2>&1
f=$(find $out -name "*_kernel_trace.csv" | head -1)
python3 - "$f" > $out/pass.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# a pass ends with k_postops (one launch: the last class workgroup merges); take the kernels between the 3rd-last and 2nd-last one of the timed loop
idx = [i for i, r in enumerate(rows) if 'k_postops' in r['Kernel_Name']]
# e2e_bench runs per-part timing passes after the loop (features x5, rpn x5): use merges only; a pass of B images ends
# with ceil(B / 8) merge launches (one per launch sequence of 8 images)
import os
m = (int(os.environ.get('BATCH', '8')) + 7) // 8
a, b = idx[-2 * m - 1] + 1, idx[-m - 1] + 1
t0 = int(rows[a]['Start_Timestamp'])
tot = 0
for r in rows[a:b]:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot += d
    print('%9.1f %8.1f  %s  grid=%s' % ((int(r['Start_Timestamp']) - t0) / 1e3, d, r['Kernel_Name'][:100], r['Grid_Size_X']))
print('kernels', b - a, 'sum of durations us', tot, 'span us', (int(rows[b - 1]['End_Timestamp']) - t0) / 1e3)
PY
find $out -name "*_kernel_trace.csv" -delete; find $out -name "*_agent_info.csv" -delete
tail -2 $out/pass.txt

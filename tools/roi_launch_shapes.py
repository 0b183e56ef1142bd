#!/usr/bin/env python3
"""Splits the k_roi_pool dispatches of a `rocprofv3 --kernel-trace` CSV by launch shape and by whether anything
else ran on the GPU at the same time: in the default bench run most B-image launches overlap the other stream
groups' kernels (their duration is a share of the machine, not a kernel property); the few that bench.py
brackets for `roofline.kernel_ms` run ALONE -- only those are comparable with it.

    python tools/roi_launch_shapes.py <kernel_trace.csv> --out profiles/<name>.json"""
import argparse, bisect, collections, csv, json

ap = argparse.ArgumentParser()
ap.add_argument('csv')
ap.add_argument('--out', required=True)
a = ap.parse_args()
rows = []
for r in csv.DictReader(open(a.csv)):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'],
                 (int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X']))) * int(r['Grid_Size_Y'])))
rows.sort()
starts = [r[0] for r in rows]
by = collections.defaultdict(list)
for i, (s, e, name, wgs) in enumerate(rows):
    if 'k_roi_pool' not in name:
        continue
    alone = True
    j = bisect.bisect_left(starts, s) - 1
    while j >= 0 and s - rows[j][0] < 5_000_000:          # earlier starts within 5 ms that are still running
        if j != i and rows[j][1] > s:
            alone = False
            break
        j -= 1
    j = bisect.bisect_right(starts, s)
    while alone and j < len(rows) and rows[j][0] < e:       # anything that starts before this one ends
        if j != i:
            alone = False
        j += 1
    by[(wgs, 'alone' if alone else 'overlapped')].append((e - s) / 1e3)
out = {'kernel': 'k_roi_pool', 'source': a.csv, 'launches': {}}
for (wgs, kind), v in sorted(by.items()):
    v.sort()
    out['launches']['%d workgroups, %s' % (wgs, kind)] = {'calls': len(v), 'avg_us': sum(v) / len(v), 'median_us': v[len(v) // 2],
                                                          'min_us': v[0], 'max_us': v[-1]}
json.dump(out, open(a.out, 'w'), indent=1)
print(json.dumps(out['launches'], indent=1))

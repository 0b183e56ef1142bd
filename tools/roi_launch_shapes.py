#!/usr/bin/env python3
"""Splits the k_roi_pool dispatches of a `rocprofv3 --kernel-trace` CSV by launch shape: the default bench run
mixes 8-image launches (8000 workgroups, overlapping with the other streams) with the few one-image launches
(1000 workgroups) that bench.py times in isolation for `roofline.kernel_ms` -- only the latter are comparable
with it.

    python tools/roi_launch_shapes.py <kernel_trace.csv> --out profiles/<name>.json"""
import argparse, collections, csv, json

ap = argparse.ArgumentParser()
ap.add_argument('csv')
ap.add_argument('--out', required=True)
a = ap.parse_args()
by = collections.defaultdict(list)
for r in csv.DictReader(open(a.csv)):
    if 'k_roi_pool' in r['Kernel_Name']:
        wgs = (int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])) * int(r['Grid_Size_Y'])
        by[wgs].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
out = {'kernel': 'k_roi_pool', 'source': a.csv, 'by_workgroups': {}}
for wgs, v in sorted(by.items()):
    v.sort()
    out['by_workgroups'][str(wgs)] = {'calls': len(v), 'avg_us': sum(v) / len(v), 'median_us': v[len(v) // 2],
                                      'min_us': v[0], 'max_us': v[-1]}
json.dump(out, open(a.out, 'w'), indent=1)
print(json.dumps(out['by_workgroups'], indent=1))

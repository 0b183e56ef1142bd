#!/usr/bin/env python3
"""Per-phase means of the counters tools/pmc_roi.sh collected: the k_roi_pool dispatches of tools/roi_bench.py come
in four phases of `reps` launches each (8-image cold, 8-image warm, one-image cold, one-image warm)."""
import argparse
import csv
import glob
import json
import os

ap = argparse.ArgumentParser()
ap.add_argument('dir')
ap.add_argument('--reps', type=int, default=8)
a = ap.parse_args()
phases = ['batch_cold', 'batch_warm', 'one_cold', 'one_warm']
out = {}
for f in sorted(glob.glob(os.path.join(a.dir, 'p*', '*', '*_counter_collection.csv'))):
    rows = [r for r in csv.DictReader(open(f)) if 'k_roi_pool' in r['Kernel_Name']]
    by_counter = {}
    for r in rows:
        by_counter.setdefault(r['Counter_Name'], []).append((int(r['Dispatch_Id']), float(r['Counter_Value'])))
    for name, v in by_counter.items():
        v.sort()
        vals = [x for _, x in v]
        d = {}
        for k, ph in enumerate(phases):
            seg = vals[k * a.reps:(k + 1) * a.reps][2:]
            if seg:
                d[ph] = sum(seg) / len(seg)
        out[name] = d
print(json.dumps(out, indent=1))

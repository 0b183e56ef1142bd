#!/usr/bin/env python3
"""Timeline statistics of a bench.py kernel trace (rocprofv3 --kernel-trace csv): how many RoI launches are in flight over
the steady-state part, how long nothing but small kernels run.   python tools/exp/bench_timeline.py <kernel_trace.csv>"""
import csv, sys
raw = list(csv.DictReader(open(sys.argv[1])))
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in raw]
grid = {r['Kernel_Name'].split('(')[0][:40]: (r['Grid_Size_X'], r['Workgroup_Size_X'], r.get('LDS_Block_Size', '?'), r.get('VGPR_Count', '?')) for r in raw}
rows.sort()
roi = [(s, e) for s, e, n in rows if 'k_roi_pool<1, 1, float, 0>' in n]
# steady state: the middle 60 % of the RoI launches
a, b = roi[len(roi) // 5][0], roi[-len(roi) // 5][1]
ev = []
for s, e in roi:
    if e > a and s < b:
        ev.append((max(s, a), 1)); ev.append((min(e, b), -1))
ev.sort()
hist, cur, last = {}, 0, a
for t, d in ev:
    hist[cur] = hist.get(cur, 0) + (t - last); last = t; cur += d
hist[cur] = hist.get(cur, 0) + (b - last)
span = b - a
n_roi = sum(1 for s, e in roi if s >= a and e <= b)
print('steady-state span %.1f ms, %d RoI launches inside -> %.1f us per 8-image group' % (span / 1e6, n_roi, span / 1e3 / max(n_roi, 1)))
for k in sorted(hist):
    print('  %d RoI launches in flight: %5.1f %% of the time' % (k, 100.0 * hist[k] / span))
import collections
agg = collections.defaultdict(lambda: [0, 0])
for s, e, n in rows:
    if s >= a and e <= b:
        k = n.split('(')[0][:40]
        agg[k][0] += 1; agg[k][1] += e - s
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]:
    print('  %-42s x%5d  avg %7.1f us  busy %5.1f %% of the span   grid %s wg %s lds %s vgpr %s' % ((k, c, t / c / 1e3, 100.0 * t / span) + grid[k]))

#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace CSV of a pipelined bench run: over the middle third of the trace, the share of
time with any kernel running, with an RoI kernel running, with two or more RoI kernels running, the average
duration of every kernel, and a 700-us excerpt of the timeline per queue."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '?')) for r in rows]
ev.sort()
t0, t1 = ev[0][0], ev[-1][1]
a, b = t0 + (t1 - t0) // 3, t0 + 2 * (t1 - t0) // 3
win = [e for e in ev if e[1] > a and e[0] < b]
def short(n):
    n = n.replace('void ', '')
    return n.split('(')[0].split('<')[0][:18]
def union(evs):
    pts = []
    for s, e, *_ in evs:
        pts.append((max(s, a), 1)); pts.append((min(e, b), -1))
    pts.sort()
    depth, last, busy1, busy2 = 0, a, 0, 0
    for t, d in pts:
        if depth >= 1: busy1 += t - last
        if depth >= 2: busy2 += t - last
        depth += d; last = t
    return busy1 / (b - a), busy2 / (b - a)
print('window %.1f ms, %d kernels' % ((b - a) / 1e6, len(win)))
print('any kernel running: %.3f   two or more: %.3f' % union(win))
roi = [e for e in win if 'k_roi_pool' in e[2]]
print('RoI kernel running: %.3f   two or more RoI kernels: %.3f   RoI launches %d, mean %.1f us' % (*union(roi), len(roi), sum(e[1] - e[0] for e in roi) / max(1, len(roi)) / 1e3))
nonroi = [e for e in win if 'k_roi_pool' not in e[2]]
print('a non-RoI kernel running: %.3f' % union(nonroi)[0])
agg = collections.defaultdict(lambda: [0, 0])
for s, e, n, q in win:
    agg[short(n)][0] += 1; agg[short(n)][1] += e - s
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print('  %-20s %5d launches  mean %7.1f us  total %.2f of the window' % (n, c, t / c / 1e3, t / (b - a)))
# excerpt
x0 = a + (b - a) // 2
qs = sorted({e[3] for e in win})
print('timeline excerpt (us from x0), per queue:')
for q in qs:
    line = []
    for s, e, n, qq in win:
        if qq == q and e > x0 and s < x0 + 700000:
            line.append('%s[%d..%d]' % (short(n)[2:9], (s - x0) // 1000, (e - x0) // 1000))
    print(' q%s: %s' % (q, ' '.join(line)))

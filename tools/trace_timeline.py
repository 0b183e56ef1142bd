#!/usr/bin/env python3
"""Prints the per-kernel timeline of ONE bench step from a rocprofv3 --kernel-trace CSV
(start offset, gap to the previous kernel, duration, name, grid, workgroup)."""
import csv
import sys


def main(path, marker='k_rp_prepare', which=-3):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    idx = [i for i, r in enumerate(rows) if marker in r['Kernel_Name']]
    a, b = idx[which], idx[which + 1]
    # a step starts with the memset that precedes the marker kernel
    while a > 0 and 'fillBuffer' in rows[a - 1]['Kernel_Name']:
        a -= 1
    while b > 0 and 'fillBuffer' in rows[b - 1]['Kernel_Name']:
        b -= 1
    t0 = int(rows[a]['Start_Timestamp'])
    prev_end = t0
    for r in rows[a:b]:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        print('%8.1f gap %6.1f dur %6.1f  %-44s grid=%s wg=%s' % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3,
              r['Kernel_Name'][:44], r['Grid_Size_X'], r['Workgroup_Size_X']))
        prev_end = e
    print('step total (us)', (int(rows[b]['Start_Timestamp']) - t0) / 1e3)


if __name__ == '__main__':
    main(sys.argv[1], *(sys.argv[2:3]))

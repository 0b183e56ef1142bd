#!/usr/bin/env python3
"""Turns two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, see
/opt/skills/guides/MI355X_MICROARCH.md "rocprofv3 PMC slots") of `bench.py` into
profiles/<name>_traffic.json: HBM-side bytes per launch of one kernel.

    python tools/pmc_traffic.py --fetch <f_counter_collection.csv> --write <w_counter_collection.csv> \
        --kernel k_roi_pool --workload <bench config key> --out profiles/roi_pool_traffic.json

Corrections (MI355X_MICROARCH.md, HBM section): both counters are in KiB; on gfx950 FETCH_SIZE
reports exactly 1/2 of the bytes of a wide coalesced read stream (16 B/lane global_load), which is
the only kind of load this kernel issues, so the read side is doubled; WRITE_SIZE is exact for
16 B/lane streaming stores.  Cross-check in the same pass: k_decode reads 2 x N x 16 B = 8.55 MB and
FETCH_SIZE reports 4.29 MB -> the 1/2 factor holds on this pool."""
import argparse
import csv
import json


def mean_counter(paths, counter, kernel_substr):
    """(round 6: bench.py runs config 5 in a child process, which the profiler follows: one counter file per process)"""
    paths = [paths] if isinstance(paths, str) else list(paths)
    vals = [float(r['Counter_Value']) for path in paths for r in csv.DictReader(open(path))
            if r['Counter_Name'] == counter and kernel_substr in r['Kernel_Name']]
    if not vals:
        raise SystemExit('no %s rows for %s in %s' % (counter, kernel_substr, paths))
    return sum(vals) / len(vals), len(vals)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--fetch', required=True, nargs='+')
    ap.add_argument('--write', required=True, nargs='+')
    ap.add_argument('--kernel', default='k_roi_pool')
    ap.add_argument('--workload', required=True)
    ap.add_argument('--images-per-launch', type=int, default=1, help='images sharing one launch in the profiled run (--batch)')
    ap.add_argument('--out', required=True)
    a = ap.parse_args()
    f_kib, nf = mean_counter(a.fetch, 'FETCH_SIZE', a.kernel)
    w_kib, nw = mean_counter(a.write, 'WRITE_SIZE', a.kernel)
    read_bytes = 2.0 * f_kib * 1024.0
    write_bytes = w_kib * 1024.0
    out = dict(workload=a.workload, kernel=a.kernel, images_per_launch=a.images_per_launch, launches_fetch=nf, launches_write=nw,
               FETCH_SIZE_KiB_per_launch=f_kib, WRITE_SIZE_KiB_per_launch=w_kib,
               read_bytes_per_launch=read_bytes, write_bytes_per_launch=write_bytes,
               hbm_bytes_per_launch=read_bytes + write_bytes,
               corrections='FETCH_SIZE x2 (gfx950 counts 128-B requests as 64 B on wide coalesced reads); '
                           'WRITE_SIZE exact; units KiB; separate --pmc passes')
    json.dump(out, open(a.out, 'w'), indent=1)
    print(json.dumps(out))


if __name__ == '__main__':
    main()

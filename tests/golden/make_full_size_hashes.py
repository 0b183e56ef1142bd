#!/usr/bin/env python3
"""Writes tests/golden/full_size_hashes.json: SHA-256 digests of the discrete outputs of the hot path at the
BASELINE.json shapes, produced by the CPU oracle (C restatement of the reference path) on the seeded synthetic
inputs of tests/full_size_cases.py.  SURVEY.md 8(c): "one full-size fixture hash per config (store SHA-256 of
kept indices, not the tensors)".

    python tests/golden/make_full_size_hashes.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import full_size_cases as fs  # noqa: E402


def main():
    out = {}
    for name in fs.CASES:
        inp = fs.make_inputs(name)
        d = fs.digests(fs.oracle_outputs(name, inp))
        d['num_anchors'] = int(inp['num_anchors'])
        out[name] = d
        print(name, d['counts'])
    with open(fs.HASH_FILE, 'w') as f:
        json.dump(out, f, indent=1, sort_keys=True)
        f.write('\n')


if __name__ == '__main__':
    main()

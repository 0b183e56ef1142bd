#!/usr/bin/env python3
"""Round 4: a detector pass of B images with its 8-image launch sequences of the hot path on side streams (parallel) or one
after the other (serial), alternating on one box.   python tools/r04/parallel_sequences_ab.py [B, default 30]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tf_eager_object_detection_amd.model.fpn_detector import ResNetFpnDetector
from tf_eager_object_detection_amd import pipeline
B = int(sys.argv[1]) if len(sys.argv) > 1 else 30
torch.manual_seed(0)
m = ResNetFpnDetector(101, 21, (800, 1333), 1000, dtype=torch.float16, max_batch=B, blind_chunks=2, batched=True).prepare()
rng = np.random.default_rng(0)
img = torch.from_numpy((rng.uniform(0, 255, (B, 800, 1333, 3)) - np.float32([103.939, 116.779, 123.68])).astype(np.float32)).cuda()
sbs = [v for v in vars(m).values() if isinstance(v, pipeline.FpnStepBatch)]
assert sbs, 'no FpnStepBatch found on the detector'
def run(n):
    for _ in range(n):
        m(img)
    torch.cuda.synchronize()
for rep in range(3):
    for par in (True, False):
        for sb in sbs:
            sb.parallel_sequences = par
        run(2)
        t0 = time.perf_counter(); run(8); el = time.perf_counter() - t0
        print('%s sequences: %.1f img/s' % ('parallel' if par else 'serial  ', 8 * B / el), flush=True)

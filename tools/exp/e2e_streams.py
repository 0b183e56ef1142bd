#!/usr/bin/env python3
"""Experiment: S detector instances of B images each on S streams, enqueued alternately by one thread (do memory-bound and
MFMA-bound layers of different streams overlap?).   python tools/exp/e2e_streams.py S B [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tf_eager_object_detection_amd.model.fpn_detector import ResNetFpnDetector
S, B = int(sys.argv[1]), int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
torch.manual_seed(0)
models = [ResNetFpnDetector(101, 21, (800, 1333), 1000, dtype=torch.float16, max_batch=B, blind_chunks=2, batched=True).prepare() for _ in range(S)]
rng = np.random.default_rng(0)
img = torch.from_numpy((rng.uniform(0, 255, (B, 800, 1333, 3)) - np.float32([103.939, 116.779, 123.68])).astype(np.float32)).cuda()
streams = [torch.cuda.Stream() for _ in range(S)]
torch.cuda.synchronize()
def run(n):
    for _ in range(n):
        for m, s in zip(models, streams):
            with torch.cuda.stream(s):
                m(img)
    torch.cuda.synchronize()
run(3)
t0 = time.perf_counter(); run(steps); el = time.perf_counter() - t0
print('%d streams x %d images: %.1f img/s' % (S, B, steps * S * B / el))

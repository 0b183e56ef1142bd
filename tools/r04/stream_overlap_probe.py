#!/usr/bin/env python3
"""Round 4: do the matrix-bound and the memory-bound launches of two half batches overlap on the CUs when the tiles are
small enough for two workgroups of DIFFERENT launches to share a CU (128 x 128 tiles: 64 KB of LDS)?  float16
ResNet-101-FPN, 800x1333; S detector instances of B images on S streams, passes enqueued alternately.

    python tools/r04/stream_overlap_probe.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tools._diag
if not os.environ.get('ODET_LIB_PATH'):
    tools._diag.use_diag_build()      # (odet_debug_* exist only in the -DODET_DIAG build: include/odet_diag.h)
import numpy as np, torch
from tf_eager_object_detection_amd import _lib
from tf_eager_object_detection_amd.model import fpn_detector as fpn

torch.manual_seed(0)
rng = np.random.default_rng(0)
FUSED = fpn._FUSED_TAIL_MIN_SLABS


def run(S, B, tile, fused, steps=6):
    for form in (0, 1):
        _lib.call('odet_debug_conv_tile', form, *(tile if tile else (0, 0, 0, 0)))
    fpn._FUSED_TAIL_MIN_SLABS = FUSED if fused else 10 ** 9
    models = [fpn.ResNetFpnDetector(101, 21, (800, 1333), 1000, dtype=torch.float16, max_batch=B, blind_chunks=2,
                                    batched=True).prepare() for _ in range(S)]
    img = torch.from_numpy((rng.uniform(0, 255, (B, 800, 1333, 3)) - np.float32([103.939, 116.779, 123.68])).astype(np.float32)).cuda()
    streams = [torch.cuda.Stream() for _ in range(S)]
    torch.cuda.synchronize()

    def go(n):
        for _ in range(n):
            for m, s in zip(models, streams):
                with torch.cuda.stream(s):
                    m(img)
        torch.cuda.synchronize()
    go(2)
    t0 = time.perf_counter(); go(steps); el = time.perf_counter() - t0
    del models
    torch.cuda.empty_cache()
    return steps * S * B / el


res = {}
for name, S, B, tile, fused in (('1 stream x 30, product', 1, 30, None, True),
                                ('2 streams x 15, product', 2, 15, None, True),
                                ('1 stream x 30, 128x128 tiles, fused tails', 1, 30, (8, 2, 2, 2), True),
                                ('2 streams x 15, 128x128 tiles, fused tails', 2, 15, (8, 2, 2, 2), True),
                                ('1 stream x 30, 128x128 tiles, unfused', 1, 30, (8, 2, 2, 2), False),
                                ('2 streams x 15, 128x128 tiles, unfused', 2, 15, (8, 2, 2, 2), False),
                                ('3 streams x 10, 128x128 tiles, unfused', 3, 10, (8, 2, 2, 2), False),
                                ('2 streams x 15, product tiles, unfused', 2, 15, None, False)):
    v = run(S, B, tile, fused)
    res[name] = round(v, 1)
    print('%-50s %8.1f img/s' % (name, v), flush=True)
for form in (0, 1):
    _lib.call('odet_debug_conv_tile', form, 0, 0, 0, 0)
os.makedirs('gpurun_out', exist_ok=True)
json.dump(res, open('gpurun_out/r04_stream_overlap.json', 'w'), indent=1)

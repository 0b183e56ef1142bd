#!/usr/bin/env python3
"""odet_conv1x1_f16 time against the pixel count (workgroup rounds): python tools/exp/conv1x1_rounds.py [K N]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tf_eager_object_detection_amd import ops
_a = [v for v in sys.argv[1:] if not v.startswith('--')]
K, N = (int(_a[0]), int(_a[1])) if len(_a) > 1 else (256, 1024)
MS = (128 * 64, 33600) if '--short' in sys.argv else (128 * 64, 128 * 128, 128 * 192, 128 * 256, 33600, 128 * 320, 128 * 384, 128 * 512, 128 * 1024)
for M in MS:
    x = torch.randn(M, K, device='cuda', dtype=torch.float16)
    w = torch.randn(N, K, device='cuda', dtype=torch.float16) * 0.05
    b = torch.randn(N, device='cuda', dtype=torch.float16)
    r = torch.randn(M, N, device='cuda', dtype=torch.float16)
    out = torch.empty_like(r)
    for res in (r, None):
        f = lambda: ops.conv1x1_f16(x, w, b, res, True, out=out)
        for _ in range(5):
            f()
        torch.cuda.synchronize()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            f()
        e.record(); torch.cuda.synchronize()
        t = a.elapsed_time(e) / 20 * 1e3
        nt = N if N <= 256 else 256
        wgs = ((M + 127) // 128) * (N // nt)
        print('M %7d  workgroups %5d  %s  %6.1f us  (%.2f us per 256 workgroups, %.0f TFLOP/s)'
              % (M, wgs, 'shortcut' if res is not None else 'no short', t, t / wgs * 256, 2.0 * M * K * N / t / 1e6))

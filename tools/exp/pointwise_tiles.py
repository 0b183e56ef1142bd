#!/usr/bin/env python3
"""Every workgroup tile of the pointwise kernel (ODET_PW_TILE=wn,mt) on the detector's layer shapes -> what the host's
tile picker should choose.   python tools/exp/pointwise_tiles.py [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tf_eager_object_detection_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8

def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / n * 1e3

rows = [('conv3 b1 sc s2', 200, 334, 256, 512, 2), ('conv3 b1 c1 s2', 200, 334, 256, 128, 2), ('conv3 c1', 100, 167, 512, 128, 1),
        ('conv4 b1 sc s2', 100, 167, 512, 1024, 2), ('conv4 b1 c1 s2', 100, 167, 512, 256, 2), ('conv4 c1', 50, 84, 1024, 256, 1),
        ('conv5 b1 sc s2', 50, 84, 1024, 2048, 2), ('conv5 b1 c1 s2', 50, 84, 1024, 512, 2), ('conv5 c1', 25, 42, 2048, 512, 1),
        ('conv5 c3', 25, 42, 512, 2048, 1), ('neck p5', 25, 42, 2048, 256, 1), ('conv2 c1', 200, 334, 256, 64, 1),
        ('l4', 50, 84, 1024, 256, 1), ('l3', 100, 167, 512, 256, 1), ('l2', 200, 334, 256, 256, 1),
        ('fc1', 1, 1000 * B // B, 12544, 1024, 1), ('fc2', 1, 1000 * B // B, 1024, 1024, 1)]
tiles = [(4, 8), (4, 7), (4, 6), (4, 5), (4, 4), (2, 4), (2, 3), (2, 2), (1, 2), (1, 1)]
print('batch %d; columns: wn,mt = channels x pixels' % B)
print('%-16s' % '' + ''.join('%9s' % ('%dx%d' % (64 * wn, (8 // wn) * 16 * mt)) for wn, mt in tiles) + '   picker')
for name, H, W, K, N, s in rows:
    x = torch.randn(B, H, W, K, device='cuda').half()
    w = (torch.randn(N, K, device='cuda') * K ** -0.5).half()
    b = torch.randn(N, device='cuda').half()
    out = torch.empty(B, (H + s - 1) // s, (W + s - 1) // s, N, device='cuda', dtype=torch.float16)
    ts = []
    for wn, mt in tiles:
        if N % (64 * wn):
            ts.append(None); continue
        os.environ['ODET_PW_TILE'] = '%d,%d' % (wn, mt)
        ts.append(timed(lambda: ops.pointwise_f16(x, w, b, None, True, s, out=out)))
    os.environ.pop('ODET_PW_TILE')
    tp = timed(lambda: ops.pointwise_f16(x, w, b, None, True, s, out=out))
    print('%-16s' % name + ''.join('%9s' % ('-' if t is None else '%.1f' % t) for t in ts) + '   %.1f' % tp)

#!/usr/bin/env python3
"""float16 implicit-GEMM kernel: every tile of its list forced through odet_debug_conv_tile on the detector's layer shapes at a
large batch (warm, back-to-back) -> does the launcher's cost model pick the fastest?   python tools/r05/f16_tiles.py [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tools._diag
if not os.environ.get('ODET_LIB_PATH'):
    tools._diag.use_diag_build()      # (odet_debug_* exist only in the -DODET_DIAG build: include/odet_diag.h)
import torch
from tf_eager_object_detection_amd import ops, _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 30
torch.manual_seed(0)


def timed(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / n * 1e3


layers = [('conv2 3x3', 200, 334, 64, 64, 3, 1), ('conv3 3x3', 100, 167, 128, 128, 3, 1), ('conv4 3x3', 50, 84, 256, 256, 3, 1),
          ('conv5 3x3', 25, 42, 512, 512, 3, 1), ('smooth P2', 200, 334, 256, 256, 3, 1), ('smooth P3', 100, 167, 256, 256, 3, 1),
          ('conv2 c1', 200, 334, 256, 64, 1, 1), ('conv3 c1', 100, 167, 512, 128, 1, 1),
          ('conv3 c3', 100, 167, 128, 512, 1, 1), ('conv4 c1', 50, 84, 1024, 256, 1, 1), ('conv4 c3', 50, 84, 256, 1024, 1, 1),
          ('conv5 c1', 25, 42, 2048, 512, 1, 1), ('conv5 c3', 25, 42, 512, 2048, 1, 1), ('conv4 b1 sc s2', 100, 167, 512, 1024, 1, 2),
          ('l2', 200, 334, 256, 256, 1, 1), ('l3', 100, 167, 512, 256, 1, 1), ('l4', 50, 84, 1024, 256, 1, 1), ('p5', 25, 42, 2048, 256, 1, 1),
          ('fc1', 1, 1000, 12544, 1024, 1, 1), ('fc2', 1, 1000, 1024, 1024, 1, 1)]
# (waves, waves along channels, 16-pixel tiles per wave, stages) of csrc/conv3x3.hip's kTiles
tiles = [(8, 4, 4, 2), (8, 4, 5, 2), (8, 4, 6, 2), (8, 4, 7, 2), (8, 4, 8, 2), (8, 2, 2, 2), (8, 2, 3, 2), (8, 2, 4, 2), (8, 1, 1, 2), (8, 1, 2, 2),
         (8, 4, 2, 2), (8, 4, 3, 2), (8, 2, 1, 2)]
for name, H, W, cin, cout, k, s in layers:
    x = torch.randn(B, H, W, cin, device='cuda').half()
    w = (torch.randn(cout, cin, k, k, device='cuda') * (cin * k * k) ** -0.5).half().contiguous(memory_format=torch.channels_last)
    bias = torch.randn(cout, device='cuda').half()
    w2 = w.reshape(cout, cin).contiguous() if k == 1 else None
    form = 0 if k == 3 else 1
    fn = (lambda: ops.conv3x3_f16(x, w, bias, relu=True)) if k == 3 else (lambda: ops.pointwise(x, w2, bias, None, True, s))
    _lib.call('odet_debug_conv_tile', form, 0, 0, 0, 0)
    pick = timed(fn)
    res = []
    for (nw, wn, mt, ns) in tiles:
        if cout % (64 * wn):
            continue
        _lib.call('odet_debug_conv_tile', form, nw, wn, mt, ns)
        try:
            res.append((timed(fn), '%dx%d' % ((nw // wn) * 16 * mt, 64 * wn)))
        except Exception:
            pass
    _lib.call('odet_debug_conv_tile', form, 0, 0, 0, 0)
    res.sort()
    gf = 2.0 * B * ((H + s - 1) // s) * ((W + s - 1) // s) * cin * cout * k * k
    flag = '' if pick <= res[0][0] * 1.03 else '   <-- pick %.0f %% behind' % (100 * (pick / res[0][0] - 1))
    print('%-15s pick %8.1f us %6.0f TF | best ' % (name, pick, gf / pick / 1e6) + ', '.join('%s %.1f' % (t[1], t[0]) for t in res[:3]) + flag, flush=True)

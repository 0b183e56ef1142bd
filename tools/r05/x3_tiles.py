#!/usr/bin/env python3
"""split-precision kernel: its tiles against each other on a few layers (odet_debug_x3_tile), correctness against the product's
pick included.  (mt, wn): (4,2) 256x128 | (2,2) 128x128 | (4,4) 128x256 | (2,1) 256x64 | (1,1) 128x64 | (8,2) wave-specialised 256x128"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tools._diag
import torch
from tf_eager_object_detection_amd import ops, _lib
torch.manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 15
layers = [('rpn P2', B, 200, 334, 256, 512, 3), ('conv4 3x3', B, 50, 84, 256, 256, 3), ('conv3 3x3', B, 100, 167, 128, 128, 3),
          ('conv4 first', B, 50, 84, 1024, 256, 1), ('conv4 last', B, 50, 84, 256, 1024, 1), ('conv3 last', B, 100, 167, 128, 512, 1),
          ('fc1', 1, 1, 1000 * B, 12544, 1024, 1)]
for name, b, H, W, cin, cout, k in layers:
    x = torch.randn(b, H, W, cin, device='cuda')
    w = torch.randn(cout, cin, k, k, device='cuda') * 0.02
    bias = torch.randn(cout, device='cuda')
    wl = w.contiguous(memory_format=torch.channels_last)
    w2 = w.reshape(cout, cin).contiguous() if k == 1 else None
    gf = 2.0 * b * H * W * cin * cout * k * k
    line = '%-12s' % name
    ref = None
    with ops.f32_form('x3'):
        fn = (lambda: ops.conv3x3_f32(x, wl, bias, relu=True)) if k == 3 else (lambda: ops.pointwise(x, w2, bias, None, True))
        for mt, wn in ((0, 0), (4, 2), (8, 2), (2, 2), (4, 4), (2, 1), (1, 1)):
            if mt and cout % (64 * wn):
                continue
            _lib.call('odet_debug_x3_tile', mt, wn)
            y = fn(); torch.cuda.synchronize()
            if ref is None:
                ref = y.clone()
            same = bool(torch.equal(y, ref))
            ts = []
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); fn(); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            line += ' | (%d,%d) %7.1f us %5.1f TF %s' % (mt, wn, min(ts), gf / min(ts) / 1e6, '' if same else 'DIFF')
    _lib.call('odet_debug_x3_tile', 0, 0)
    print(line, flush=True)

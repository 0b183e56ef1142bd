#!/usr/bin/env python3
"""split-precision kernel: its tiles against each other on a few layers (odet_debug_x3_tile), correctness against the product's
pick included.  (mt, wn): (4,2) 256x128 | (2,2) 128x128 | (4,4) 128x256 | (2,1) 256x64 | (1,1) 128x64 | (8,2) wave-specialised 256x128"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tools._diag
if not os.environ.get('ODET_LIB_PATH'):
    tools._diag.use_diag_build()      # (odet_debug_* exist only in the -DODET_DIAG build: include/odet_diag.h)
import torch
from tf_eager_object_detection_amd import ops, _lib
torch.manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 15
SPLITS = [int(v) for v in sys.argv[2].split(',')] if len(sys.argv) > 2 else [1]       # K splits to try per tile
FORM = sys.argv[3] if len(sys.argv) > 3 else 'x3'
layers = [('rpn P2', B, 200, 334, 256, 512, 3), ('conv4 3x3', B, 50, 84, 256, 256, 3), ('conv3 3x3', B, 100, 167, 128, 128, 3),
          ('conv5 3x3', B, 25, 42, 512, 512, 3), ('conv4 first', B, 50, 84, 1024, 256, 1), ('conv4 last', B, 50, 84, 256, 1024, 1),
          ('conv5 first', B, 25, 42, 2048, 512, 1), ('conv3 last', B, 100, 167, 128, 512, 1), ('fc1', 1, 1, 1000 * B, 12544, 1024, 1),
          ('fc2', 1, 1, 1000 * B, 1024, 1024, 1)]
for name, b, H, W, cin, cout, k in layers:
    x = torch.randn(b, H, W, cin, device='cuda')
    w = torch.randn(cout, cin, k, k, device='cuda') * 0.02
    bias = torch.randn(cout, device='cuda')
    wl = w.contiguous(memory_format=torch.channels_last)
    w2 = w.reshape(cout, cin).contiguous() if k == 1 else None
    gf = 2.0 * b * H * W * cin * cout * k * k
    line = '%-12s' % name
    ref = None
    with ops.f32_form(FORM):
        fn = (lambda: ops.conv3x3_f32(x, wl, bias, relu=True)) if k == 3 else (lambda: ops.pointwise(x, w2, bias, None, True))
        for mt, wn, ks in [(0, 0, 0)] + [(m_, w_, s_) for (m_, w_) in ((4, 2), (2, 2), (4, 4), (2, 1), (1, 1)) for s_ in SPLITS]:
            if mt and (cout % (64 * wn) or (FORM == 'x2' and (mt, wn) == (4, 2)) or (ks > 1 and (cin * k * k // 32) // ks < 4)):
                continue
            _lib.call('odet_debug_x3_tile', mt, wn, ks)
            y = fn(); torch.cuda.synchronize()
            if ref is None:
                ref = y.clone()
            same = bool(torch.equal(y, ref))
            ts = []
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); fn(); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            err = float((y - ref).abs().max()) / max(float(ref.abs().max()), 1e-30)
            line += ' | (%d,%d)x%d %7.1f us %5.1f TF%s' % (mt, wn, ks, min(ts), gf / min(ts) / 1e6, '' if same else ' d=%.1e' % err)
    _lib.call('odet_debug_x3_tile', 0, 0, 0)
    print(line, flush=True)

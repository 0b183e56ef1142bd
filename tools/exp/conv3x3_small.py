#!/usr/bin/env python3
"""64 -> 64 and 128 -> 128 3x3 layers (ResNet conv2 / conv3) on the own kernel's small tiles against the library."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from tf_eager_object_detection_amd import ops
torch.backends.cudnn.benchmark = True
def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / n * 1e3
for B in (8, 15):
    for (H, W, C) in ((200, 334, 64), (100, 167, 128)):
        x = torch.randn(B, H, W, C, device='cuda').half()
        w = (torch.randn(C, C, 3, 3, device='cuda') * 0.05).half().contiguous(memory_format=torch.channels_last)
        b = torch.randn(C, device='cuda').half()
        xn = x.permute(0, 3, 1, 2)
        t_lib = timed(lambda: F.conv2d(xn, w, None, 1, 1))
        res = []
        for tile in ('', '1,1', '1,2', '2,2', '2,3', '2,4'):
            if tile:
                os.environ['ODET_C3_TILE'] = tile
            else:
                os.environ.pop('ODET_C3_TILE', None)
            try:
                res.append((tile or 'picker', timed(lambda: ops.conv3x3_f16(x, w, b, relu=True))))
            except Exception as e:
                res.append((tile, None))
        os.environ.pop('ODET_C3_TILE', None)
        err = (ops.conv3x3_f16(x, w, None).float() - F.conv2d(xn, w, None, 1, 1).permute(0, 2, 3, 1).float()).abs().max().item()
        print('B %d %dx%dx%d  library %.1f us | ' % (B, H, W, C, t_lib) + '  '.join('%s %s' % (t, '-' if v is None else '%.1f' % v) for t, v in res) + ' | max diff %.3f' % err)

#!/usr/bin/env python3
"""64-channel-tile 3x3 layers under taller pixel tiles (ODET_C3_TILE=1,mt: TM = 128 mt).  python tools/exp/tall64_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tf_eager_object_detection_amd import ops
def timed(fn, n=8):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    e.record(); e.synchronize()
    return a.elapsed_time(e) / n * 1e3
for name, B, H, W, cin, cout in (('VGG16 conv1_2', 32, 600, 800, 64, 64), ('ResNet conv2 3x3', 30, 200, 334, 64, 64)):
    x = torch.randn(B, H, W, cin, device='cuda').half()
    w = (torch.randn(cout, cin, 3, 3, device='cuda') * 0.05).half().contiguous(memory_format=torch.channels_last)
    b = torch.randn(cout, device='cuda').half()
    out = torch.empty(B, H, W, cout, device='cuda', dtype=torch.float16)
    ref = None
    row = []
    for tile in ('', '1,1', '1,2', '1,3', '1,4'):
        if tile: os.environ['ODET_C3_TILE'] = tile
        else: os.environ.pop('ODET_C3_TILE', None)
        t = timed(lambda: ops.conv3x3_f16(x, w, b, relu=True, out=out))
        if ref is None: ref = out.clone()
        row.append('%s:%.0f%s' % (tile or 'pick', t, '' if torch.equal(out, ref) else '(DIFFERS)'))
    os.environ.pop('ODET_C3_TILE', None)
    print('%-18s %s' % (name, ' '.join(row)))

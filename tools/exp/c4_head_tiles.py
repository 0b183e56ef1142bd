#!/usr/bin/env python3
"""The C4 RoI head's 1x1 layers (conv5 on 9000 RoIs x 7 x 7 = 441 000 pixels) under every workgroup tile of the pointwise
kernel (ODET_PW_TILE), with / without the shortcut.   python tools/exp/c4_head_tiles.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tf_eager_object_detection_amd import ops
M = 9000 * 49
def timed(fn, n=6):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / n * 1e3
for name, K, N, res in (('c3 512->2048 + shortcut', 512, 2048, True), ('c1 2048->512', 2048, 512, False), ('c1 1024->512', 1024, 512, False)):
    x = torch.randn(1, 1, M, K, device='cuda').half()
    w = (torch.randn(N, K, device='cuda') * K ** -0.5).half()
    b = torch.randn(N, device='cuda').half()
    r = torch.randn(1, 1, M, N, device='cuda').half() if res else None
    out = torch.empty(1, 1, M, N, device='cuda', dtype=torch.float16)
    row = []
    for tile in ('', '4,8', '4,6', '4,4', '2,8', '2,4', '2,2', '1,4', '1,2'):
        if tile: os.environ['ODET_PW_TILE'] = tile
        else: os.environ.pop('ODET_PW_TILE', None)
        try:
            row.append('%s:%.0f' % (tile or 'pick', timed(lambda: ops.pointwise(x, w, b, r, True, 1, out=out))))
        except Exception as ex:
            row.append('%s:err' % tile)
    os.environ.pop('ODET_PW_TILE', None)
    gf = 2.0 * M * K * N / 1e9
    print('%-26s %s   (%.0f GFLOP)' % (name, ' '.join(row), gf))

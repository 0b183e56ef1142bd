#!/usr/bin/env python3
"""The fused bottleneck tail (3x3 + last 1x1 + shortcut + ReLU in one launch) against the two launches, ResNet conv2 / conv3 /
conv4 shapes.   python tools/exp/block_tail_layers.py [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tf_eager_object_detection_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 30
def timed(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / n * 1e3
for name, H, W, cm in (('conv2', 200, 334, 64), ('conv3', 100, 167, 128), ('conv4', 50, 84, 256)):
    n3 = 4 * cm
    x = torch.randn(B, H, W, cm, device='cuda').half()
    w2 = (torch.randn(cm, cm, 3, 3, device='cuda') * 0.05).half().contiguous(memory_format=torch.channels_last)
    b2 = torch.randn(cm, device='cuda').half()
    w3 = (torch.randn(n3, cm, device='cuda') * 0.05).half()
    b3 = torch.randn(n3, device='cuda').half()
    r = torch.randn(B, H, W, n3, device='cuda').half()
    out = torch.empty_like(r)
    y2 = torch.empty(B, H, W, cm, device='cuda', dtype=torch.float16)
    res = {}
    for tile in ([''] + (['4,8', '4,6', '4,5', '4,4'] if cm == 256 else ['2,4', '2,3', '2,2'] if cm == 128 else ['1,2', '1,1'])):
        if tile: os.environ['ODET_C3_TILE'] = tile
        else: os.environ.pop('ODET_C3_TILE', None)
        res[tile or 'pick'] = timed(lambda: ops.conv3x3_conv1x1_f16(x, w2, b2, w3, b3, residual=r, relu=True, out=out))
    os.environ.pop('ODET_C3_TILE', None)
    res['no-shortcut'] = timed(lambda: ops.conv3x3_conv1x1_f16(x, w2, b2, w3, b3, residual=None, relu=True, out=out))
    res['n3=64'] = timed(lambda: ops.conv3x3_conv1x1_f16(x, w2, b2, w3[:64].contiguous(), b3[:64].contiguous(), residual=None, relu=True, out=y2 if cm == 64 else None))
    def two():
        ops.conv3x3_f16(x, w2, out=y2)
        if cm <= 512:
            return ops.conv1x1_f16(y2, w3, b3, residual=r, relu=True, out=out, in_bias=b2)
    t2 = timed(two)
    t_c2 = timed(lambda: ops.conv3x3_f16(x, w2, out=y2))
    print('%s batch %d: fused %s | two launches %.1f us (3x3 alone %.1f)' % (name, B, ' '.join('%s:%.0f' % kv for kv in res.items()), t2, t_c2))

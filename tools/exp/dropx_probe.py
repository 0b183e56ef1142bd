#!/usr/bin/env python3
"""Upper bound of re-using one staged pixel window for the three dx taps: 3x3 layers with 64 / 128 input channels, product
build vs a build that drops two of three pixel copies (-DODET_C3_DROPX: wrong results).  python tools/exp/dropx_probe.py"""
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
for name, B, H, W, cin, cout in (('VGG16 conv1_2', 32, 600, 800, 64, 64), ('VGG16 conv2_1', 32, 300, 400, 64, 128), ('VGG16 conv2_2', 32, 300, 400, 128, 128),
                                 ('ResNet conv2 3x3', 30, 200, 334, 64, 64), ('ResNet conv3 3x3', 30, 100, 167, 128, 128), ('neck P2 3x3', 30, 200, 334, 256, 256)):
    x = torch.randn(B, H, W, cin, device='cuda').half()
    w = (torch.randn(cout, cin, 3, 3, device='cuda') * 0.05).half().contiguous(memory_format=torch.channels_last)
    b = torch.randn(cout, device='cuda').half()
    out = torch.empty(B, H, W, cout, device='cuda', dtype=torch.float16)
    print('%-18s %7.1f us' % (name, timed(lambda: ops.conv3x3_f16(x, w, b, relu=True, out=out))))

#!/usr/bin/env python3
"""VGG16's first convolution alone (odet_conv3x3_rgb_f16), 32 images of 600 x 800.   python tools/exp/rgb_conv_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tf_eager_object_detection_amd import ops
B, H, W = 32, 600, 800
img = torch.randn(B, H, W, 3, device='cuda') * 50
w = (torch.randn(64, 3, 3, 3, device='cuda') * 0.05).half()
b = torch.randn(64, device='cuda').half()
pw = ops.conv3x3_rgb_pack_weights(w)
out = torch.empty(B, H, W, 64, device='cuda', dtype=torch.float16)
def timed(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    e.record(); e.synchronize()
    return a.elapsed_time(e) / n * 1e3
t = timed(lambda: ops.conv3x3_rgb(img, pw, b, out=out))
print('float32 image: %.1f us  (%.2f TB/s of output)' % (t, out.numel() * 2 / t / 1e6))
img16 = img.half()
t = timed(lambda: ops.conv3x3_rgb(img16, pw, b, out=out))
print('float16 image: %.1f us  (%.2f TB/s of output)' % (t, out.numel() * 2 / t / 1e6))

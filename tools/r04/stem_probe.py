#!/usr/bin/env python3
"""Round 4: the stem kernel alone (float16 output, float32 800x1333 images) for rocprofv3 --pmc passes / timing.
    python tools/r04/stem_probe.py [batch, default 8]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tf_eager_object_detection_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
img = torch.rand(B, 800, 1333, 3, device='cuda') * 255 - 120
w = (torch.randn(64, 3, 7, 7, device='cuda') * 0.05).half()
b = torch.randn(64, device='cuda').half()
pw = ops.stem_pack_weights(w)
for _ in range(3):
    y = ops.stem_conv7_pool3(img, pw, b)
torch.cuda.synchronize()
a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10):
    y = ops.stem_conv7_pool3(img, pw, b)
e.record(); e.synchronize()
print('batch %d: %.1f us per launch' % (B, a.elapsed_time(e) * 100))

#!/usr/bin/env python3
"""one layer of tools/r05/x3_layers.py, a few launches of one form: the target of rocprofv3 --pmc / --kernel-trace runs"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tf_eager_object_detection_amd import ops
ap = argparse.ArgumentParser()
ap.add_argument('--form', default='x3')
ap.add_argument('--shape', default='15,200,334,256,512,3')
ap.add_argument('--reps', type=int, default=4)
a = ap.parse_args()
B, H, W, cin, cout, k = (int(v) for v in a.shape.split(','))
torch.manual_seed(0)
x = torch.randn(B, H, W, cin, device='cuda')
w = torch.randn(cout, cin, k, k, device='cuda') * (2.0 / (cin * k * k)) ** 0.5
b = torch.randn(cout, device='cuda') * 0.1
wl = w.contiguous(memory_format=torch.channels_last)
w2 = w.reshape(cout, cin).contiguous() if k == 1 else None
with ops.f32_form(a.form):
    for _ in range(a.reps):
        y = ops.conv3x3_f32(x, wl, b, relu=True) if k == 3 else ops.pointwise(x, w2, b, None, True)
torch.cuda.synchronize()
print('ok', float(y.abs().mean()))

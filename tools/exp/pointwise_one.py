#!/usr/bin/env python3
"""One pointwise layer shape in a loop (for counter passes): python tools/exp/pointwise_one.py H W K N stride batch [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tf_eager_object_detection_amd import ops
H, W, K, N, s, B = (int(v) for v in sys.argv[1:7])
reps = int(sys.argv[7]) if len(sys.argv) > 7 else 8
x = torch.randn(B, H, W, K, device='cuda').half()
w = (torch.randn(N, K, device='cuda') * K ** -0.5).half()
b = torch.randn(N, device='cuda').half()
out = torch.empty(B, (H + s - 1) // s, (W + s - 1) // s, N, device='cuda', dtype=torch.float16)
for _ in range(reps):
    ops.pointwise(x, w, b, None, True, s, out=out)
torch.cuda.synchronize()

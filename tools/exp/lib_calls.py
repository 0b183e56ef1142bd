#!/usr/bin/env python3
"""Which torch convolution / GEMM entry points a detector still calls in one pass, with shapes.
    python tools/exp/lib_calls.py c4|vgg16|fpn [fp16|fp32]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, torch.nn.functional as F
from tf_eager_object_detection_amd.model.fpn_detector import ResNetFpnDetector
from tf_eager_object_detection_amd.model.frcnn_detector import ResNetC4Detector, Vgg16Detector
kind = sys.argv[1] if len(sys.argv) > 1 else 'c4'
dt = torch.float32 if (len(sys.argv) > 2 and sys.argv[2] == 'fp32') else torch.float16
torch.manual_seed(0)
if kind == 'c4':
    m, shape = ResNetC4Detector(50, 21, (800, 1333), 300, dtype=dt, max_batch=2, blind_chunks=4).prepare(), (800, 1333)
elif kind == 'vgg16':
    m, shape = Vgg16Detector(21, (600, 800), 300, dtype=dt, max_batch=2, blind_chunks=4).prepare(), (600, 800)
else:
    m, shape = ResNetFpnDetector(101, 21, (800, 1333), 1000, dtype=dt, max_batch=2).prepare(), (800, 1333)
img = torch.randn((2,) + shape + (3,), device='cuda') * 50
m(img)
calls = []
def wrap(mod, name):
    real = getattr(mod, name)
    def f(*a, **k):
        calls.append((name, [tuple(t.shape) for t in a if torch.is_tensor(t)][:3], {q: v for q, v in k.items() if not torch.is_tensor(v)}, [v for v in a if isinstance(v, (int, tuple))]))
        return real(*a, **k)
    setattr(mod, name, f)
for n in ('conv2d', 'linear'):
    wrap(F, n)
for n in ('addmm', '_addmm_activation', 'matmul', 'mm'):
    wrap(torch, n)
m(img)
for c in calls:
    print(c)
print(len(calls), 'library calls')

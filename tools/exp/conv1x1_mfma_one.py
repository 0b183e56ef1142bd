#!/usr/bin/env python3
"""One layer shape through odet_conv1x1_f16 a few times (for rocprofv3 --pmc passes):
    python tools/exp/conv1x1_mfma_one.py H W CIN COUT [reps]"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tf_eager_object_detection_amd import _lib
h, w, cin, cout = (int(v) for v in sys.argv[1:5])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 10
fn = _lib.lib().odet_conv1x1_f16
fn.restype = C.c_int
fn.argtypes = [C.c_void_p] * 6 + [C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_void_p]
B = 8
x = torch.randn(B, h, w, cin, device='cuda', dtype=torch.float16)
wt = torch.randn(cout, cin, device='cuda', dtype=torch.float16) * 0.05
bias = torch.randn(cout, device='cuda', dtype=torch.float16)
res = torch.randn(B, h, w, cout, device='cuda', dtype=torch.float16)
out = torch.empty_like(res)
for _ in range(reps):
    _lib.check(fn(x.data_ptr(), None, wt.data_ptr(), bias.data_ptr(), res.data_ptr(), out.data_ptr(), B * h * w, cin, cout, 1, _lib.stream()))
torch.cuda.synchronize()
print('done')

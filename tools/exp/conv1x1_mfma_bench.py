#!/usr/bin/env python3
"""Checks and times the MFMA 1x1 convolutions -- the product's odet_conv1x1_f16 (csrc/conv1x1.hip) and, when the loaded
library has it, the experimental direct-to-register odet_conv1x1_f16_direct (tools/exp/conv1x1_mfma.hip) -- against
the library route (convolution + ops.bias_act_ with the shortcut).  For the experimental kernel:

    bash tools/exp/conv1x1_mfma_build.sh && ODET_LIB_PATH=tools/exp/_ablate/libodet_hip_conv1x1.so \\
        python tools/exp/conv1x1_mfma_bench.py"""
import ctypes as C, os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tools._diag            # (ODET_LIB_PATH selects a diagnostic build: tools only, the product reads no environment)
from tf_eager_object_detection_amd import _lib, ops

lib = _lib.lib()
name = 'odet_conv1x1_f16_direct' if (hasattr(lib, 'odet_conv1x1_f16_direct') and '--product' not in sys.argv) else 'odet_conv1x1_f16'
print('kernel:', name)
fn = getattr(lib, name)
fn.restype = C.c_int
product = name == 'odet_conv1x1_f16'          # (the product entry point has an in_bias pointer after x)
fn.argtypes = [C.c_void_p] * (6 if product else 5) + [C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_void_p]


def conv1x1(x, w, b, res, relu=True, out=None):
    cin, cout = x.shape[-1], w.shape[0]
    out = torch.empty(tuple(x.shape[:-1]) + (cout,), dtype=torch.float16, device='cuda') if out is None else out
    _lib.check(fn(x.data_ptr(), *([None] if product else []), w.data_ptr(), b.data_ptr(), res.data_ptr() if res is not None else None, out.data_ptr(),
                  x.numel() // cin, cin, cout, 1 if relu else 0, _lib.stream()))
    return out


def timeit(f, warm=5, reps=20):
    for _ in range(warm):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


g = torch.Generator(device='cuda'); g.manual_seed(1)
ri = lambda lo, hi, *sh: torch.randint(lo, hi, sh, device='cuda', generator=g).to(torch.float16)
for M, K, N in ((1007, 64, 256), (96, 128, 512), (33, 256, 1024), (2500, 256, 64), (64, 64, 320), (5000, 128, 128)):
    x, w, b, r = ri(-3, 4, M, K), ri(-2, 3, N, K), ri(-8, 9, N), ri(-16, 17, M, N)
    w[:, 0] += torch.arange(N, device='cuda').remainder(5).to(torch.float16)
    x[:, 1] += torch.arange(M, device='cuda').remainder(7).to(torch.float16)
    for res, relu in ((r, True), (None, True), (r, False)):
        want = x.float() @ w.float().t() + b.float()
        want = want + res.float() if res is not None else want
        want = torch.relu(want) if relu else want
        assert '--no-check' in sys.argv or torch.equal(conv1x1(x, w, b, res, relu).float(), want), (M, K, N)
print('exact on integer data')
torch.backends.cudnn.benchmark = True
B = 8
SHAPES = ((200, 334, 64, 256), (100, 167, 128, 512), (50, 84, 256, 1024), (200, 334, 256, 64))
if product:
    SHAPES += ((25, 42, 512, 2048), (100, 167, 512, 128), (120, 123, 512, 2048))   # conv5 / conv3 c1 / the C4 RoI head (2400 crops of 7x7 = 14760 px x 8)
for h, w_, cin, cout in SHAPES:
    x = torch.randn(B, cin, h, w_, device='cuda', dtype=torch.float16).contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(cout, cin, 1, 1, device='cuda', dtype=torch.float16) * 0.05).contiguous(memory_format=torch.channels_last)
    bias = torch.randn(cout, device='cuda', dtype=torch.float16)
    res = torch.randn(B, h, w_, cout, device='cuda', dtype=torch.float16)
    xn, w2 = x.permute(0, 2, 3, 1), wt.view(cout, cin)

    def lib_route():
        y = F.conv2d(x, wt, None)
        ops.bias_act_(y.permute(0, 2, 3, 1), bias, res, True)
        return y

    out = torch.empty_like(res)
    ta, tb = timeit(lib_route), timeit(lambda: conv1x1(xn, w2, bias, res, True, out))
    tc = timeit(lambda: conv1x1(xn, w2, bias, None, True, out))
    nb = B * h * w_ * (cin + 2 * cout) * 2
    print('%3dx%-3d %4d -> %-4d with shortcut: library %6.1f us   mfma %6.1f us (%.2f TB/s of %d MB)   | no shortcut: mfma %6.1f us'
          % (h, w_, cin, cout, ta, tb, nb / tb / 1e6, nb // 1000000, tc))

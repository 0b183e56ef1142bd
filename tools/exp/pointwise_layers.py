#!/usr/bin/env python3
"""The pointwise form of the implicit-GEMM kernel (ops.pointwise_f16 / lateral_merge_f16) on the ResNet-101-FPN layer
shapes it takes over from the library, against the library route each layer used (hipBLASLt GEMM with its own epilogue
for stride 1, the library convolution + fused epilogue pass for stride 2, GEMM + merge launch for the laterals).
    python tools/exp/pointwise_layers.py [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from tf_eager_object_detection_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
torch.backends.cudnn.benchmark = True

def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / n * 1e3

flush = torch.empty(1 << 27, dtype=torch.float32, device='cuda')
rows = [('conv3 b1 shortcut s2', 200, 334, 256, 512, 2, False), ('conv3 b1 c1 s2', 200, 334, 256, 128, 2, True),
        ('conv3 c1', 100, 167, 512, 128, 1, True), ('conv4 b1 shortcut s2', 100, 167, 512, 1024, 2, False),
        ('conv4 b1 c1 s2', 100, 167, 512, 256, 2, True), ('conv4 c1', 50, 84, 1024, 256, 1, True),
        ('conv5 b1 shortcut s2', 50, 84, 1024, 2048, 2, False), ('conv5 b1 c1 s2', 50, 84, 1024, 512, 2, True),
        ('conv5 c1', 25, 42, 2048, 512, 1, True), ('conv5 c3 (+res)', 25, 42, 512, 2048, 1, True),
        ('neck p5', 25, 42, 2048, 256, 1, False), ('conv2 b2 c1', 200, 334, 256, 64, 1, True)]
print('batch %d' % B)
for name, H, W, K, N, s, relu in rows:
    x = torch.randn(B, H, W, K, device='cuda').half()
    w = (torch.randn(N, K, device='cuda') * K ** -0.5).half()
    b = torch.randn(N, device='cuda').half()
    Ho, Wo = (H + s - 1) // s, (W + s - 1) // s
    out = torch.empty(B, Ho, Wo, N, device='cuda', dtype=torch.float16)
    res = torch.randn(B, Ho, Wo, N, device='cuda').half() if '+res' in name else None
    t_own = timed(lambda: ops.pointwise_f16(x, w, b, res, relu, s, out=out))
    xn = x.permute(0, 3, 1, 2)
    w4 = w.view(N, K, 1, 1).contiguous(memory_format=torch.channels_last)
    if s == 1 and res is None:
        x2 = x.view(-1, K)
        lib = lambda: (torch._addmm_activation(b, x2, w.t(), use_gelu=False) if relu else torch.addmm(b, x2, w.t()))
    else:
        def lib():
            y = F.conv2d(xn, w4, None, s)
            ops.bias_act_(y.permute(0, 2, 3, 1), b, res, relu)
            return y
    t_lib = timed(lib)
    ref = lib()
    ref = ref.view(B, Ho, Wo, N) if ref.dim() == 2 else ref.permute(0, 2, 3, 1)
    err = (ops.pointwise_f16(x, w, b, res, relu, s).float() - ref.float()).abs().max().item()
    gf = 2.0 * B * Ho * Wo * K * N / 1e9
    mb = (B * Ho * Wo * (K + N * (2 if res is not None else 1))) * 2 / 1e6
    print('%-22s M %7d K %5d N %5d  own %7.1f us (%5.0f TFLOP/s, %4.2f TB/s)  library %7.1f us   max|diff| %.3f'
          % (name, B * Ho * Wo, K, N, t_own, gf / t_own * 1e3, mb / t_own, t_lib, err))
# laterals with the merge
for name, H, W, K in (('l4 + merge', 50, 84, 1024), ('l3 + merge', 100, 167, 512), ('l2 + merge', 200, 334, 256)):
    N = 256
    x = torch.randn(B, H, W, K, device='cuda').half()
    w = (torch.randn(N, K, device='cuda') * K ** -0.5).half()
    b = torch.randn(N, device='cuda').half()
    top = torch.randn(B, (H + 1) // 2, (W + 1) // 2, N, device='cuda').half()
    out = torch.empty(B, H, W, N, device='cuda', dtype=torch.float16)
    t_own = timed(lambda: ops.lateral_merge_f16(x, w, b, top, out=out))
    x2 = x.view(-1, K)
    def lib():
        lat = torch.addmm(b, x2, w.t()).view(B, H, W, N)
        return ops.fpn_topdown_merge(top, lat)
    t_lib = timed(lib)
    err = (ops.lateral_merge_f16(x, w, b, top).float() - lib().float()).abs().max().item()
    print('%-22s M %7d K %5d N %5d  own %7.1f us  GEMM + merge launch %7.1f us   max|diff| %.3f' % (name, B * H * W, K, N, t_own, t_lib, err))
# RoI head
for name, M, K, N in (('fc1', 1000 * B, 12544, 1024), ('fc2', 1000 * B, 1024, 1024)):
    x = torch.randn(M, K, device='cuda').half()
    w = (torch.randn(N, K, device='cuda') * K ** -0.5).half()
    b = torch.randn(N, device='cuda').half()
    out = torch.empty(M, N, device='cuda', dtype=torch.float16)
    t_own = timed(lambda: ops.dense_f16(x, w, b, True, out=out))
    t_lib = timed(lambda: torch._addmm_activation(b, x, w.t(), use_gelu=False))
    err = (ops.dense_f16(x, w, b, True).float() - torch._addmm_activation(b, x, w.t(), use_gelu=False).float()).abs().max().item()
    print('%-22s M %7d K %5d N %5d  own %7.1f us (%5.0f TF/s)  library %7.1f us   max|diff| %.3f'
          % (name, M, K, N, t_own, 2.0 * M * K * N / t_own * 1e-6, t_lib, err))

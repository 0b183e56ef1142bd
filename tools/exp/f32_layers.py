#!/usr/bin/env python3
"""The float32 forms (conv_f32.hip) on the ResNet-101-FPN layer shapes against the library routes (MIOpen / hipBLASLt, find
mode).   python tools/exp/f32_layers.py [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from tf_eager_object_detection_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
torch.backends.cudnn.benchmark = True
torch.backends.cuda.matmul.allow_tf32 = False

def timed(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / n * 1e3

print('batch %d' % B)
rows = [('conv2 c1 256->64', 200, 334, 256, 64, 1), ('conv2 c3 64->256', 200, 334, 64, 256, 1), ('conv3 c1 512->128', 100, 167, 512, 128, 1),
        ('conv3 c3 128->512', 100, 167, 128, 512, 1), ('conv3 b1 c1 s2', 200, 334, 256, 128, 2), ('conv4 c1 1024->256', 50, 84, 1024, 256, 1),
        ('conv4 c3 256->1024', 50, 84, 256, 1024, 1), ('conv5 c1 2048->512', 25, 42, 2048, 512, 1), ('conv5 c3 512->2048', 25, 42, 512, 2048, 1),
        ('l2 256->256', 200, 334, 256, 256, 1), ('fc1', 1, 1000, 12544, 1024, 1), ('rpn 1x1 P2 512->64', 200, 334, 512, 64, 1)]
for name, H, W, K, N, s in rows:
    x = torch.randn(B, H, W, K, device='cuda')
    w = torch.randn(N, K, device='cuda') * K ** -0.5
    b = torch.randn(N, device='cuda')
    out = torch.empty(B, (H + s - 1) // s, (W + s - 1) // s, N, device='cuda')
    ts = {}
    for tile in ('', '4,8', '4,6', '4,4', '2,4', '2,2', '1,2', '1,1'):
        if tile:
            os.environ['ODET_F32_TILE'] = tile
        else:
            os.environ.pop('ODET_F32_TILE', None)
        try:
            ts[tile or 'pick'] = timed(lambda: ops.pointwise(x, w, b, None, True, s, out=out))
        except Exception:
            pass
    os.environ.pop('ODET_F32_TILE', None)
    xn = x.permute(0, 3, 1, 2); w4 = w.view(N, K, 1, 1).contiguous(memory_format=torch.channels_last)
    if s == 1:
        x2 = x.view(-1, K)
        t_lib = timed(lambda: torch._addmm_activation(b, x2, w.t(), use_gelu=False))
    else:
        def lib():
            y = F.conv2d(xn, w4, None, s); ops.bias_act_(y.permute(0, 2, 3, 1), b, None, True); return y
        t_lib = timed(lib)
    gf = 2.0 * (out.numel() // N) * K * N / 1e9
    print('%-22s own %s | library %.1f us | %.0f TF/s own-pick' % (name, ' '.join('%s:%.0f' % kv for kv in ts.items()), t_lib, gf / ts['pick'] * 1e3))
for name, H, W, C, N in (('conv2 3x3 64', 200, 334, 64, 64), ('conv3 3x3 128', 100, 167, 128, 128), ('conv4 3x3 256', 50, 84, 256, 256),
                         ('conv5 3x3 512', 25, 42, 512, 512), ('neck s2 256', 200, 334, 256, 256)):
    x = torch.randn(B, H, W, C, device='cuda')
    w = (torch.randn(N, C, 3, 3, device='cuda') * 0.05).contiguous(memory_format=torch.channels_last)
    b = torch.randn(N, device='cuda')
    ts = {}
    for tile in ('', '4,8', '4,4', '2,4', '2,2', '1,2', '1,1'):
        if tile:
            os.environ['ODET_F32_TILE'] = tile
        else:
            os.environ.pop('ODET_F32_TILE', None)
        try:
            ts[tile or 'pick'] = timed(lambda: ops.conv3x3_f32(x, w, b, relu=True))
        except Exception:
            pass
    os.environ.pop('ODET_F32_TILE', None)
    xn = x.permute(0, 3, 1, 2)
    t_lib = timed(lambda: F.conv2d(xn, w, None, 1, 1))
    gf = 2.0 * B * H * W * C * 9 * N / 1e9
    print('%-22s own %s | library %.1f us | %.0f TF/s own-pick' % (name, ' '.join('%s:%.0f' % kv for kv in ts.items()), t_lib, gf / ts['pick'] * 1e3))

#!/usr/bin/env python3
"""Round 6: is the float16 3x3 loop limited by the matrix pipe's clock under load (DVFS on random operands) or by its structure?
The product kernel and the probe (and its bare-MFMA variant) on the RpnHead's P2 level with RANDOM and with ZERO operands."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tf_eager_object_detection_amd import ops
HERE = os.path.dirname(os.path.abspath(__file__))
libs = {}
for v in ('', '_NODMA_NOREAD_NOSYNC'):
    l = C.CDLL(os.path.join(HERE, 'libconv_v2_probe%s.so' % v))
    l.v2_conv3x3_f16.restype = C.c_int
    l.v2_conv3x3_f16.argtypes = [C.c_void_p] * 4 + [C.c_int] * 6 + [C.c_void_p]
    libs[v or 'v2'] = l


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return min(ts)


B, H, W, cin, cout = 15, 200, 334, 256, 512
gf = 2.0 * B * H * W * cin * cout * 9
for kind in ('normal', 'uniform[-1,1)', 'zeros', 'ones'):
    if kind == 'normal':
        x = torch.randn((B, H, W, cin), device='cuda').half(); w = (torch.randn((cout, cin, 3, 3), device='cuda') * 0.03).half()
    elif kind.startswith('uniform'):
        x = (torch.rand((B, H, W, cin), device='cuda') * 2 - 1).half(); w = (torch.rand((cout, cin, 3, 3), device='cuda') * 2 - 1).half()
    elif kind == 'zeros':
        x = torch.zeros((B, H, W, cin), device='cuda').half(); w = torch.zeros((cout, cin, 3, 3), device='cuda').half()
    else:
        x = torch.ones((B, H, W, cin), device='cuda').half(); w = torch.ones((cout, cin, 3, 3), device='cuda').half()
    w = w.contiguous(memory_format=torch.channels_last)
    y = torch.empty((B, H, W, cout), dtype=torch.float16, device='cuda')
    wk = w.permute(0, 2, 3, 1)
    out = ['%-14s' % kind]
    t = timed(lambda: ops.conv3x3_f16(x, w, None, relu=True))
    out.append('product %7.1f us %7.1f TF' % (t, gf / t / 1e6))
    for name, l in libs.items():
        t = timed(lambda: l.v2_conv3x3_f16(x.data_ptr(), wk.data_ptr(), None, y.data_ptr(), B, H, W, cin, cout, 1, torch.cuda.current_stream().cuda_stream))
        out.append('%s %7.1f us %7.1f TF' % (name, t, gf / t / 1e6))
    print(' | '.join(out), flush=True)

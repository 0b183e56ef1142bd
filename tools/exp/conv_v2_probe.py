#!/usr/bin/env python3
"""Round 6 probe (tools/exp/conv_v2_probe.hip): the one-wave-per-SIMD 3x3 loop with a 128 x 128 wave tile in the accumulator file
against the product's k_conv3x3_f16 -- exactness on integer data first, then times on the RpnHead's P2 level and conv5's 3x3.
    hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/exp/conv_v2_probe.hip -o tools/exp/libconv_v2_probe.so   (build container)
    python tools/exp/conv_v2_probe.py [batch]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from tf_eager_object_detection_amd import ops

VARIANT = os.environ.get('V2_VARIANT', '')
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libconv_v2_probe%s.so' % (('_' + VARIANT) if VARIANT else '')))
lib.v2_conv3x3_f16.restype = C.c_int
lib.v2_conv3x3_f16.argtypes = [C.c_void_p] * 4 + [C.c_int] * 6 + [C.c_void_p]


def v2(x, w, b, relu=True):
    B, H, W, cin = x.shape
    cout = w.shape[0]
    y = torch.empty((B, H, W, cout), dtype=torch.float16, device=x.device)
    wk = w.permute(0, 2, 3, 1)
    assert wk.is_contiguous()
    rc = lib.v2_conv3x3_f16(x.data_ptr(), wk.data_ptr(), b.data_ptr() if b is not None else None, y.data_ptr(), B, H, W, cin, cout,
                            1 if relu else 0, torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc
    return y


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return min(ts)


g = torch.Generator(device='cuda'); g.manual_seed(3)
# exactness on integer data (every product and sum exact in float32 / float16), borders, partial tiles, K tails of 1..3 steps
for (B, H, W, cin, cout) in ((1, 5, 7, 32, 256), (2, 33, 47, 64, 256), (1, 40, 56, 96, 512), (3, 19, 23, 256, 256)):
    xi = torch.randint(-2, 3, (B, H, W, cin), device='cuda', generator=g).half()
    wi = ((torch.randint(0, 100, (cout, cin, 3, 3), device='cuda', generator=g) < 15).half()
          * torch.randint(-2, 3, (cout, cin, 3, 3), device='cuda', generator=g).half()).contiguous(memory_format=torch.channels_last)
    bi = torch.randint(-3, 4, (cout,), device='cuda', generator=g).half()
    want = torch.relu(F.conv2d(xi.permute(0, 3, 1, 2).float(), wi.float(), bi.float(), 1, 1)).permute(0, 2, 3, 1)
    got = v2(xi, wi, bi)
    ok = torch.equal(got.float(), want)
    print('exact on integers %s: %s' % ((B, H, W, cin, cout), ok), flush=True)
    assert ok or VARIANT
# random data against the product kernel (same float32 accumulation, another order: float16 rounding apart)
Bt = int(sys.argv[1]) if len(sys.argv) > 1 else 15
for name, (B, H, W, cin, cout) in (('rpn P2', (Bt, 200, 334, 256, 512)), ('smooth P2', (Bt, 200, 334, 256, 256)), ('conv4 3x3', (Bt, 50, 84, 256, 256)),
                                   ('conv5 3x3', (Bt, 25, 42, 512, 512))):
    x = torch.randn((B, H, W, cin), device='cuda', generator=g).half()
    w = (torch.randn((cout, cin, 3, 3), device='cuda', generator=g) * (2.0 / (9 * cin)) ** 0.5).half().contiguous(memory_format=torch.channels_last)
    b = (torch.randn(cout, device='cuda', generator=g) * 0.1).half()
    ref = ops.conv3x3_f16(x, w, b, relu=True)
    got = v2(x, w, b)
    err = float((got.float() - ref.float()).abs().max()) / float(ref.float().abs().max())
    t_ref = timed(lambda: ops.conv3x3_f16(x, w, b, relu=True))
    t_v2 = timed(lambda: v2(x, w, b))
    gf = 2.0 * B * H * W * cin * cout * 9
    print('%-10s %s: product %8.1f us %7.1f TF | v2 %8.1f us %7.1f TF | max rel diff %.2e' % (name, (B, H, W, cin, cout), t_ref, gf / t_ref / 1e6,
                                                                                               t_v2, gf / t_v2 / 1e6, err), flush=True)

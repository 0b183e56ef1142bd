#!/usr/bin/env python3
"""the float16 split-K experiment (tools/exp/conv3x3_f16_split_k.patch, see f16_splitk.py) on the layers where the library's GEMM
is ahead of the product's pick (profiles/r05_pointwise_vs_library.txt): warm, back-to-back launches, against torch's GEMM.
Needs the patched working tree + variant library, as f16_splitk.py says.

Round 5 (us; tile = waves, waves along channels, 16-pixel tiles per wave, stages, K split):
    fc1 b1        library  51.8 | product pick  63.9 | best (8,2,2,2) x4 56.5, ring 8 stages 58.2
    fc1 b4        library 104.1 | product pick 149.0 | best (8,4,4,2) x2 112.7
    conv5 c1 b4   library  22.0 | product pick  24.7 | best (8,1,1,2) x1 22.6
    conv5 c1 b30  library  53.9 | product pick  61.4 | best (8,4,8,2) x2 59.5
    fc2 b30       library  55.8 | product pick  61.9 | best 61.5        p5 b30   library 34.6 | product pick 35.1
With a warm L2 the K split does help the deepest layer (fc1: -12 % / -24 %) without reaching the library; fc1 is 2.7 % of a
batch-4 pass.  Not adopted (a stream-bound workspace for < 1 % of a pass); the cold-L2 sweep of f16_splitk.py found no gain."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tools._diag
if not os.environ.get('ODET_LIB_PATH'):
    tools._diag.use_diag_build()      # (odet_debug_* exist only in the -DODET_DIAG build: include/odet_diag.h)
import torch
from tf_eager_object_detection_amd import ops, _lib
torch.manual_seed(0)
lib = _lib.lib()
ws = torch.zeros(int(lib.odet_conv_workspace_bytes()), dtype=torch.uint8, device='cuda')
lib.odet_conv_workspace_bind(torch.cuda.current_stream().cuda_stream, ws.data_ptr(), ws.numel())


def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / n * 1e3


layers = [('fc1 b1', 1000, 12544, 1024), ('fc1 b4', 4000, 12544, 1024), ('conv5 c1 b4', 4200, 2048, 512), ('conv5 c1 b30', 31500, 2048, 512),
          ('fc2 b30', 30000, 1024, 1024), ('p5 b30', 31500, 2048, 256)]
tiles = [(0, 0, 0, 0), (4, 1, 1, 4), (4, 1, 1, 8), (8, 2, 2, 2), (8, 1, 1, 2), (8, 4, 4, 2), (8, 2, 4, 2), (8, 4, 8, 2)]
for name, M, K, N in layers:
    x = torch.randn(1, 1, M, K, device='cuda').half()
    w = (torch.randn(N, K, device='cuda') * K ** -0.5).half()
    b = torch.randn(N, device='cuda').half()
    out = torch.empty(1, 1, M, N, device='cuda', dtype=torch.float16)
    fn = lambda: ops.pointwise(x, w, b, None, True, out=out)
    x2 = x.view(M, K)
    t_lib = timed(lambda: torch._addmm_activation(b, x2, w.t(), use_gelu=False))
    res = []
    for (nw, wn, mt, ns) in tiles:
        if nw and N % (64 * wn):
            continue
        for S in ((0,) if nw == 0 else (1, 2, 3, 4, 6, 8)):
            if S > 1 and (K // 64) // S < 2:
                continue
            _lib.call('odet_debug_conv_tile', 1, nw, wn, mt, ns)
            _lib.call('odet_debug_conv_split', 1, S)
            try:
                res.append((timed(fn), (nw, wn, mt, ns, S)))
            except Exception as ex:
                pass
    _lib.call('odet_debug_conv_tile', 1, 0, 0, 0, 0); _lib.call('odet_debug_conv_split', 1, 0)
    pick = res[0][0]
    best = sorted(res)[:3]
    print('%-13s library %6.1f us | product pick %6.1f | best ' % (name, t_lib, pick) + '  '.join('%s %.1f' % (t[1], t[0]) for t in best), flush=True)

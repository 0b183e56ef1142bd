#!/usr/bin/env python3
"""float16 implicit-GEMM kernel: tile x K-split sweep on the small-map layers at batch 1 / 2 / 4 (odet_debug_conv_tile /
odet_debug_conv_split), each launch inside a HIP graph with a cold L2 (as tools/r04/small_tiles.py measures).

Needs a library built from tools/exp/conv3x3_f16_split_k.patch (split-K for the plain 3x3 / pointwise float16 forms: S workgroups
per tile, float32 parts in a stream-bound workspace, fixed-order reduction by the last one -- the form csrc/conv_x3.hip has):
    python -c "import tools._diag as d; d.build_variant('tools/exp/libodet_f16_splitk.so', patch='tools/exp/conv3x3_f16_split_k.patch', only=['conv3x3.hip'])"
    ODET_LIB_PATH=tools/exp/libodet_f16_splitk.so python tools/r05/f16_splitk.py 1      (the patched _lib.py / odet.h are needed too)

Round 5, batch 1, us per launch (graph replay, cold L2), product pick vs the best split of any tile -- BUILT, CORRECT (exact on
integers, deterministic), and REJECTED: the ring forms the launcher picks are not beaten anywhere.
    conv4 3x3   ring 64x64/4 stages 32.6 | best split 128x128 x2 41.0        conv5 3x3   40.2 | ring 8 stages 39.6, ring4 x2 41.5
    conv4 first ring 24.2              | 128x64 x1 28.9                       conv5 first 26.2 | ring4 25.3, 128x64 x2 35.8
    conv4 last  21.0                   | 128x128 x1 21.0                      p5          25.4 | ring4 x2 25.9
    fc1         81.4                   | 128x128 x6 84.3                      fc2         21.0 | ring4 20.2
The float16 small-map layers are bound by a CU's intake from L2, which the ring already saturates with 64 x 64 tiles on every
CU; larger tiles + split-K trade half the operand bytes for the parts' traffic and a second hand-off and come out behind."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tools._diag
if not os.environ.get('ODET_LIB_PATH'):
    tools._diag.use_diag_build()      # (odet_debug_* exist only in the -DODET_DIAG build: include/odet_diag.h)
import torch
from tf_eager_object_detection_amd import ops, _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
torch.manual_seed(0)
lib = _lib.lib()
ws = torch.zeros(int(lib.odet_conv_workspace_bytes()), dtype=torch.uint8, device='cuda')
flush = torch.empty(1 << 27, dtype=torch.float32, device='cuda')
layers = [('conv4 3x3', B, 50, 84, 256, 256, 3), ('conv4 first', B, 50, 84, 1024, 256, 1), ('conv4 last', B, 50, 84, 256, 1024, 1),
          ('conv5 3x3', B, 25, 42, 512, 512, 3), ('conv5 first', B, 25, 42, 2048, 512, 1), ('conv5 last', B, 25, 42, 512, 2048, 1),
          ('conv3 3x3', B, 100, 167, 128, 128, 3), ('p5', B, 25, 42, 2048, 256, 1), ('fc1', 1, 1, 1000 * B, 12544, 1024, 1), ('fc2', 1, 1, 1000 * B, 1024, 1024, 1)]
tiles = [(0, 0, 0, 0), (4, 1, 1, 4), (4, 1, 1, 8), (8, 2, 2, 2), (8, 1, 1, 2), (8, 4, 4, 2), (8, 2, 4, 2)]
for name, b, H, W, cin, cout, k in layers:
    x = torch.randn(b, H, W, cin, device='cuda').half()
    w = (torch.randn(cout, cin, k, k, device='cuda') * (cin * k * k) ** -0.5).half().contiguous(memory_format=torch.channels_last)
    bias = torch.randn(cout, device='cuda').half()
    w2 = w.reshape(cout, cin).contiguous() if k == 1 else None
    out = torch.empty(b, H, W, cout, device='cuda', dtype=torch.float16)
    form = 0 if k == 3 else 1
    fn = (lambda: ops.conv3x3_f16(x, w, bias, relu=True, out=out)) if k == 3 else (lambda: ops.pointwise(x, w2, bias, None, True, out=out))
    st = torch.cuda.Stream()
    lib.odet_conv_workspace_bind(st.cuda_stream, ws.data_ptr(), ws.numel())
    line, ref = '%-11s' % name, None
    for (nw, wn, mt, ns) in tiles:
        if nw and cout % (64 * wn):
            continue
        for S in ((0,) if nw == 0 else (1, 2, 3, 4, 6, 8)):
            if S > 1 and (cin * k * k // 64) // S < 2:
                continue
            _lib.call('odet_debug_conv_tile', form, nw, wn, mt, ns)
            _lib.call('odet_debug_conv_split', form, S)
            with torch.cuda.stream(st):
                fn(); st.synchronize()
                y = out.float().clone()
                if ref is None:
                    ref = y
                err = float((y - ref).abs().max()) / max(float(ref.abs().max()), 1e-9)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=st):
                    fn()
                ts = []
                for _ in range(6):
                    flush.fill_(1.0)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(st); g.replay(); e1.record(st); st.synchronize()
                    ts.append(e0.elapsed_time(e1) * 1e3)
            line += ' | %d,%d,%d,%dx%d %5.1f%s' % (nw, wn, mt, ns, S, sorted(ts)[1], '' if err < 2e-3 else ' ERR%.0e' % err)
    _lib.call('odet_debug_conv_tile', form, 0, 0, 0, 0); _lib.call('odet_debug_conv_split', form, 0)
    print(line, flush=True)

#!/usr/bin/env python3
"""Round 4: a stage's first bottleneck as one contraction (ops.pointwise_dual) -- K = cin1 + cin2 is small (128 for conv2), so
a launch is prologue + epilogue: which tile serves it best at 30 images?  Also the other small-K pointwise layers of conv2.
    python tools/r04/dual_tiles.py [batch, default 30]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tools._diag
if not os.environ.get('ODET_LIB_PATH'):
    tools._diag.use_diag_build()      # (odet_debug_* exist only in the -DODET_DIAG build: include/odet_diag.h)
import torch
from tf_eager_object_detection_amd import ops, _lib


def timed(fns, n=8):
    """us per call inside a HIP graph of n calls cycling through the closures (each on its own tensors: cold L2)"""
    for f in fns:
        f()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fns[0]()
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for i in range(n):
                fns[i % len(fns)]()
    torch.cuda.current_stream().wait_stream(side)
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(3):
        g.replay()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / (3 * n) * 1e3


def force(form, tile):
    _lib.call('odet_debug_conv_tile', form, *(tile if tile else (0, 0, 0, 0)))


B = int(sys.argv[1]) if len(sys.argv) > 1 else 30
TILES = [None, (8, 4, 8, 2), (8, 4, 4, 2), (8, 4, 2, 2), (8, 2, 4, 2), (8, 2, 2, 2), (8, 2, 1, 2), (8, 1, 2, 2), (8, 1, 1, 2)]
res = {}
for name, H, W, c1, c2, cout, stride in (('conv2 first block: [c2 out 64 | stem 64] -> 256', 200, 334, 64, 64, 256, 1),
                                         ('conv3 first block: [128 | conv2 256, stride 2] -> 512', 200, 334, 128, 256, 512, 2),
                                         ('conv4 first block: [256 | conv3 512, stride 2] -> 1024', 100, 167, 256, 512, 1024, 2)):
    Ho, Wo = (H + stride - 1) // stride, (W + stride - 1) // stride
    sets = []
    for _ in range(2):
        sets.append((torch.randn(B, Ho, Wo, c1, device='cuda').half(), torch.randn(B, H, W, c2, device='cuda').half(),
                     torch.empty(B, Ho, Wo, cout, device='cuda', dtype=torch.float16)))
    w = (torch.randn(cout, c1 + c2, device='cuda') * 0.05).half()
    b = torch.randn(cout, device='cuda').half()
    row = {}
    ref = None
    for tile in TILES:
        if tile and cout % (64 * tile[1]):
            continue
        force(1, tile)
        try:
            y = ops.pointwise_dual(sets[0][0], sets[0][1], w, b, stride, True).clone()
            torch.cuda.synchronize()
            if ref is None:
                ref = y
            same = bool(torch.equal(y, ref))
            t = timed([(lambda s_=s_: ops.pointwise_dual(s_[0], s_[1], w, b, stride, True, out=s_[2])) for s_ in sets], n=8)
            row['pick' if tile is None else '%d,%d,%d,%d' % tile] = [round(t, 1), same]
        except Exception as ex:
            row[str(tile)] = 'error: %s' % ex
    force(1, None)
    res[name] = row
    print(name, json.dumps(row), flush=True)
os.makedirs('gpurun_out', exist_ok=True)
json.dump({'batch': B, 'us': res}, open('gpurun_out/r04_dual_tiles.json', 'w'), indent=1)

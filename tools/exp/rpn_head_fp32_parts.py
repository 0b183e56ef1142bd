"""Where the float32 RpnHead's time goes at batch 4 (800x1333 pyramid): the grouped 3x3 convolution, the 1x1 pair
through the library, the pack pass."""
import json, os, sys
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tf_eager_object_detection_amd import ops
torch.backends.cudnn.benchmark = True
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
A = 3
def timed(fn, n=10):
    for _ in range(4):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / n
shapes = [(200, 334), (100, 167), (50, 84), (25, 42), (13, 21)]
xs = [torch.randn(B, h, w, 256, device='cuda') for h, w in shapes]
w3 = (torch.randn(512, 256, 3, 3, device='cuda') * 0.01).contiguous(memory_format=torch.channels_last)
b3 = torch.randn(512, device='cuda')
w1 = (torch.randn(6 * A, 512, 1, 1, device='cuda') * 0.01).contiguous(memory_format=torch.channels_last)
b1 = torch.randn(6 * A, device='cuda')
n = sum(h * w for h, w in shapes) * A
scores = torch.empty((B, n, 2), device='cuda'); deltas = torch.empty((B, n, 4), device='cuda')
heads = ops.conv3x3_f32_levels(xs, w3, b3, relu=True)
res = {}
res['conv3x3_levels_ms'] = timed(lambda: ops.conv3x3_f32_levels(xs, w3, b3, relu=True))
def pair():
    return [F.conv2d(h.permute(0, 3, 1, 2), w1, None).permute(0, 2, 3, 1) for h in heads]
res['conv1x1_pair_library_ms'] = timed(pair)
sds = pair()
res['pair_contiguous'] = [bool(s.is_contiguous()) for s in sds]
def pack():
    off = 0
    for (h, w), sd in zip(shapes, sds):
        ops.rpn_pack_pair(sd if sd.is_contiguous() else sd.contiguous(), b1, A, scores, deltas, off)
        off += h * w * A
res['pack_ms'] = timed(pack)
for i, hd in enumerate(heads):
    res['conv1x1_level%d_ms' % i] = timed(lambda: F.conv2d(hd.permute(0, 3, 1, 2), w1, None))
mm = heads[0].reshape(-1, 512)
res['matmul_level0_ms'] = timed(lambda: mm @ w1.reshape(18, 512).t())
print(json.dumps(res))

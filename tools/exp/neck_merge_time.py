"""How long the three top-down merges of the FPN neck (TF1 legacy bilinear resize + 0.5/0.5 fusion, torch ops)
take inside the detector: decides whether a fused HIP kernel is worth writing."""
import sys, torch, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tf_eager_object_detection_amd.model.fpn_detector import tf_legacy_resize_bilinear
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dt = torch.float16
shapes = [(200, 334), (100, 167), (50, 84), (25, 42)]
mk = lambda hw: torch.randn(B, 256, hw[0], hw[1], device='cuda', dtype=dt).contiguous(memory_format=torch.channels_last)
p5, l4, l3, l2 = mk(shapes[3]), mk(shapes[2]), mk(shapes[1]), mk(shapes[0])
def merges():
    p4 = tf_legacy_resize_bilinear(p5, shapes[2]) * 0.5 + l4 * 0.5
    p3 = tf_legacy_resize_bilinear(p4, shapes[1]) * 0.5 + l3 * 0.5
    p2 = tf_legacy_resize_bilinear(p3, shapes[0]) * 0.5 + l2 * 0.5
    return p2
for _ in range(5): merges()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): merges()
b.record(); torch.cuda.synchronize()
print('three merges (torch ops), batch %d: %.3f ms per batch' % (B, a.elapsed_time(b) / 20))
from tf_eager_object_detection_amd import ops
nh = lambda t: t.permute(0, 2, 3, 1)
def fused():
    p4 = ops.fpn_topdown_merge(nh(p5), nh(l4))
    p3 = ops.fpn_topdown_merge(p4, nh(l3))
    return ops.fpn_topdown_merge(p3, nh(l2))
for _ in range(5): fused()
torch.cuda.synchronize()
a.record()
for _ in range(20): fused()
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / 20
byt = sum(2 * B * h * w * 256 * 2 for h, w in shapes[:3])
print('three merges (odet_fpn_topdown_merge), batch %d: %.3f ms per batch = %.2f TB/s of lateral-in + out bytes' % (B, ms, byt / ms / 1e9))

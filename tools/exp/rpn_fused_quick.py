"""The RpnHead at batch 8 (800x1333 pyramid): fused launch vs convolution launch + five tail launches."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tf_eager_object_detection_amd import ops
B, A = 8, 3
def timed(fn, n=30):
    for _ in range(8): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / n
shapes = [(200, 334), (100, 167), (50, 84), (25, 42), (13, 21)]
xs = [torch.randn(B, h, w, 256, device='cuda').half() for h, w in shapes]
w3 = (torch.randn(512, 256, 3, 3, device='cuda') * 0.01).half().contiguous(memory_format=torch.channels_last)
b3 = torch.randn(512, device='cuda').half()
w1 = (torch.randn(6 * A, 512, device='cuda') * 0.05).half()
b1 = torch.randn(6 * A, device='cuda').half()
n = sum(h * w for h, w in shapes) * A
scores = torch.empty((B, n, 2), device='cuda'); deltas = torch.empty((B, n, 4), device='cuda')
def two_pass():
    convs = ops.conv3x3_f16_levels(xs, w3)
    off = 0
    for (h, w), c in zip(shapes, convs):
        ops.rpn_head_tail(c, b3, w1, b1, A, scores, deltas, off)
        off += h * w * A
print('conv levels alone  %.3f ms' % timed(lambda: ops.conv3x3_f16_levels(xs, w3)))
print('two-pass head      %.3f ms' % timed(two_pass))
print('fused head         %.3f ms' % timed(lambda: ops.rpn_head_fused(xs, w3, b3, w1, b1, A, scores, deltas)))

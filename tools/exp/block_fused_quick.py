"""A conv4 bottleneck's 3x3 + last 1x1 at batch 8 (50x84, 256 -> 256 -> 1024): fused launch vs the two launches."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tf_eager_object_detection_amd import ops
def timed(fn, n=40):
    for _ in range(8): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / n * 1e3
for B in (8, 4, 1):
    H, W = 50, 84
    x = torch.randn(B, H, W, 256, device='cuda').half()
    w2 = (torch.randn(256, 256, 3, 3, device='cuda') * 0.01).half().contiguous(memory_format=torch.channels_last)
    b2 = torch.randn(256, device='cuda').half()
    w3 = (torch.randn(1024, 256, device='cuda') * 0.02).half()
    b3 = torch.randn(1024, device='cuda').half()
    r = torch.randn(B, H, W, 1024, device='cuda').half()
    out = torch.empty_like(r); mid = torch.empty(B, H, W, 256, device='cuda', dtype=torch.float16)
    t3 = timed(lambda: ops.conv3x3_f16(x, w2, out=mid))
    t1 = timed(lambda: ops.conv1x1_f16(mid, w3, b3, residual=r, relu=True, out=out, in_bias=b2))
    tf = timed(lambda: ops.conv3x3_conv1x1_f16(x, w2, b2, w3, b3, residual=r, relu=True, out=out))
    print('batch %d: conv3x3 %.1f + conv1x1 %.1f = %.1f us;  fused %.1f us' % (B, t3, t1, t3 + t1, tf))

"""odet_conv1x1_f16 (+ shortcut + ReLU) on the bottleneck shapes of ResNet-101 at batch 8: quick A/B of builds."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tf_eager_object_detection_amd import ops
def timed(fn, n=30):
    for _ in range(8): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / n * 1e3
B = 8
for name, h, w, cin, cout, res in (('conv4_c3', 50, 84, 256, 1024, True), ('conv3_c3', 100, 167, 128, 512, True), ('conv2_c3', 200, 334, 64, 256, True),
                                   ('conv2_c1', 200, 334, 256, 64, False), ('conv5_c3', 25, 42, 512, 2048, True)):
    x = torch.randn(B, h, w, cin, device='cuda').half()
    wt = (torch.randn(cout, cin, device='cuda') * 0.02).half()
    bias = torch.randn(cout, device='cuda').half()
    r = torch.randn(B, h, w, cout, device='cuda').half() if res else None
    out = torch.empty(B, h, w, cout, device='cuda', dtype=torch.float16)
    t = timed(lambda: ops.conv1x1_f16(x, wt, bias, residual=r, relu=True, out=out))
    mb = (x.numel() + (r.numel() if res else 0) + out.numel()) * 2 / 1e6
    print('%-9s %7.1f us  %6.0f MB  %5.2f TB/s' % (name, t, mb, mb / t))

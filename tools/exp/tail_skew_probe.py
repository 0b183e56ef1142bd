import os, sys
sys.path.insert(0, '/root/repo')
import torch
from tf_eager_object_detection_amd import ops
B=30
def timed(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / n * 1e3
H,W,cm=50,84,256
n3=1024
x = torch.randn(B, H, W, cm, device='cuda').half()
w2 = (torch.randn(cm, cm, 3, 3, device='cuda') * 0.05).half().contiguous(memory_format=torch.channels_last)
b2 = torch.randn(cm, device='cuda').half()
w3 = (torch.randn(n3, cm, device='cuda') * 0.05).half()
b3 = torch.randn(n3, device='cuda').half()
r = torch.randn(B, H, W, n3, device='cuda').half()
out = torch.empty_like(r)
for sk in (0, 2, 5, 8, 11, 14, 18):
    os.environ['ODET_C3_SKEW'] = str(sk)
    print('skew %2d x 3.4 us: %.1f us' % (sk, timed(lambda: ops.conv3x3_conv1x1_f16(x, w2, b2, w3, b3, residual=r, relu=True, out=out))))

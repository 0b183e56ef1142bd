"""The ResNet stem at batch 8 / 4 / 1 (800x1333): fused launch vs cast + library convolution + bias/ReLU/pool pass."""
import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tf_eager_object_detection_amd import ops
torch.backends.cudnn.benchmark = True
def timed(fn, n=30):
    for _ in range(6): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / n * 1e3
w = (torch.randn(64, 3, 7, 7, device='cuda') * 0.02).half()
wl = w.contiguous(memory_format=torch.channels_last)
b = torch.randn(64, device='cuda').half()
pw = ops.stem_pack_weights(w)
for B in (8, 4, 1):
    img = torch.randn(B, 800, 1333, 3, device='cuda') * 50
    def lib():
        x = img.to(torch.float16).permute(0, 3, 1, 2)
        y = F.conv2d(x, wl, None, 2, 3)
        if not y.is_contiguous(memory_format=torch.channels_last):
            y = y.contiguous(memory_format=torch.channels_last)
        return ops.bias_relu_maxpool(y.permute(0, 2, 3, 1), b, 3, 2, 1, False)
    print('batch %d: library route %.1f us, fused stem %.1f us' % (B, timed(lib), timed(lambda: ops.stem_conv7_pool3(img, pw, b))))

"""odet_conv3x3_f16 against the library convolution (MIOpen, find mode) on the RPN head's five level shapes at batch 8."""
import json, os, sys
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tf_eager_object_detection_amd import ops
torch.backends.cudnn.benchmark = True
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
w = (torch.randn(512, 256, 3, 3, device='cuda') * 0.01).half().contiguous(memory_format=torch.channels_last)
res, tot_l, tot_o = {}, 0.0, 0.0
def timed(fn, n=20):
    for _ in range(5):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / n
for name, (h, wd) in (('P2', (200, 334)), ('P3', (100, 167)), ('P4', (50, 84)), ('P5', (25, 42)), ('P6', (13, 21))):
    x = torch.randn(B, h, wd, 256, device='cuda').half()
    xc = x.permute(0, 3, 1, 2)
    lib = timed(lambda: F.conv2d(xc, w, None, 1, 1))
    own = timed(lambda: ops.conv3x3_f16(x, w))
    fl = 2.0 * B * h * wd * 512 * 2304
    err = (ops.conv3x3_f16(x, w).float() - F.conv2d(xc, w, None, 1, 1).permute(0, 2, 3, 1).float()).abs().max().item()
    res[name] = dict(library_ms=lib, own_ms=own, library_tflops=fl / lib / 1e9, own_tflops=fl / own / 1e9, max_abs_diff=err)
    tot_l += lib; tot_o += own
res['total_library_ms'], res['total_own_ms'] = tot_l, tot_o
print(json.dumps(res))

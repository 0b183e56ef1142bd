"""Library (MIOpen, find mode) time of the RPN head's 3x3 convolution 256 -> 512 per pyramid level at batch 8, float16
NHWC -- the number a hand-written implicit-GEMM kernel has to beat (base_fpn_model.py:401-417)."""
import torch, torch.nn.functional as F, json
torch.backends.cudnn.benchmark = True
B = 8
res = {}
w = (torch.randn(512, 256, 3, 3, device='cuda') * 0.01).half().contiguous(memory_format=torch.channels_last)
tot = 0.0
for name, (h, wd) in (('P2', (200, 334)), ('P3', (100, 167)), ('P4', (50, 84)), ('P5', (25, 42)), ('P6', (13, 21))):
    x = torch.randn(B, 256, h, wd, device='cuda').half().contiguous(memory_format=torch.channels_last)
    for _ in range(5):
        y = F.conv2d(x, w, None, 1, 1)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        y = F.conv2d(x, w, None, 1, 1)
    b.record(); b.synchronize()
    ms = a.elapsed_time(b) / 20
    fl = 2.0 * B * h * wd * 512 * 2304
    res[name] = dict(ms=ms, tflops=fl / ms / 1e9)
    tot += ms
res['total_ms'] = tot
print(json.dumps(res))

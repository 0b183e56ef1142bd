"""odet_conv3x3_f16 alone on the RPN head's P2 / P3 shapes and the grouped launch (batch 8): a quick A/B of kernel edits."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tf_eager_object_detection_amd import ops
B = 8
w = (torch.randn(512, 256, 3, 3, device='cuda') * 0.01).half().contiguous(memory_format=torch.channels_last)
def timed(fn, n=30):
    for _ in range(8):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / n
shapes = [(200, 334), (100, 167), (50, 84), (25, 42), (13, 21)]
xs = [torch.randn(B, h, wd, 256, device='cuda').half() for h, wd in shapes]
res = {}
for rep in range(2):
    res['P2_ms_%d' % rep] = round(timed(lambda: ops.conv3x3_f16(xs[0], w)), 4)
    res['P3_ms_%d' % rep] = round(timed(lambda: ops.conv3x3_f16(xs[1], w)), 4)
    res['levels_ms_%d' % rep] = round(timed(lambda: ops.conv3x3_f16_levels(xs, w)), 4)
fl = 2.0 * B * 200 * 334 * 512 * 2304
res['P2_tflops'] = round(fl / min(res['P2_ms_0'], res['P2_ms_1']) / 1e9)
print(json.dumps(res))

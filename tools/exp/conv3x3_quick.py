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
# mid-size layers (where the pixel-tile height matters): ResNet conv4 c2 / neck s4, neck s3, C4 RPN, VGG
for name, (h, wd, cin, cout) in (('conv4_c2', (50, 84, 256, 256)), ('neck_s3', (100, 167, 256, 256)), ('neck_s2', (200, 334, 256, 256)),
                                 ('c4_rpn', (50, 84, 1024, 512)), ('vgg_conv4', (75, 100, 512, 512)), ('conv5_c2', (25, 42, 512, 512))):
    x = torch.randn(B, h, wd, cin, device='cuda').half()
    ww = (torch.randn(cout, cin, 3, 3, device='cuda') * 0.01).half().contiguous(memory_format=torch.channels_last)
    t = timed(lambda: ops.conv3x3_f16(x, ww))
    print(name, round(t * 1e3, 1), 'us', round(2.0 * B * h * wd * cout * 9 * cin / t / 1e9), 'TFLOP/s')

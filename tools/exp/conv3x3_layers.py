"""odet_conv3x3_f16 / _f32 (argv[2] = f32) against the library convolution on every eligible 3x3 layer shape of the detectors at batch 8."""
import json, os, sys
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tf_eager_object_detection_amd import ops
torch.backends.cudnn.benchmark = True
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
DT = torch.float32 if len(sys.argv) > 2 and sys.argv[2] == "f32" else torch.float16
own_fn = ops.conv3x3_f32 if DT == torch.float32 else ops.conv3x3_f16
def timed(fn, n=20):
    for _ in range(5):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / n
res = {}
for name, h, wd, cin, cout in (('rpn_p2', 200, 334, 256, 512), ('rpn_p3', 100, 167, 256, 512), ('conv2_c2 (x3)', 200, 334, 64, 64), ('conv3_c2 (x4)', 100, 167, 128, 128), ('neck_s2', 200, 334, 256, 256), ('neck_s3', 100, 167, 256, 256), ('neck_s4', 50, 84, 256, 256),
                               ('conv4_c2 (x23)', 50, 84, 256, 256), ('conv5_c2 (x3)', 25, 42, 512, 512),
                               ('c4_rpn', 50, 84, 1024, 512), ('vgg_rpn 600x800', 38, 50, 512, 512),
                               ('vgg conv4 600x800', 75, 100, 512, 512), ('vgg conv3 600x800', 150, 200, 256, 256)):
    x = torch.randn(B, h, wd, cin, device='cuda').to(DT)
    w = (torch.randn(cout, cin, 3, 3, device='cuda') * 0.01).to(DT).contiguous(memory_format=torch.channels_last)
    xc = x.permute(0, 3, 1, 2)
    lib = timed(lambda: F.conv2d(xc, w, None, 1, 1))
    own = timed(lambda: own_fn(x, w))
    fl = 2.0 * B * h * wd * cout * 9 * cin
    res[name] = dict(library_ms=round(lib, 4), own_ms=round(own, 4), library_tflops=round(fl / lib / 1e9), own_tflops=round(fl / own / 1e9))
print(json.dumps(res))

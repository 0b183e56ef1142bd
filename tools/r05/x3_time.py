import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tools._diag
if not os.environ.get('ODET_LIB_PATH'):
    tools._diag.use_diag_build()      # (odet_debug_* exist only in the -DODET_DIAG build: include/odet_diag.h)
import torch
from tf_eager_object_detection_amd import ops
torch.manual_seed(0)
from tf_eager_object_detection_amd import _lib
FORM = sys.argv[4] if len(sys.argv) > 4 else 'x3'
if len(sys.argv) > 2:
    _lib.call('odet_debug_x3_tile', int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 1)
for (B, H, W, cin, cout, k) in ((15, 200, 334, 256, 512, 3), (15, 1, 1000, 12544, 1024, 1)):
    x = torch.randn(B, H, W, cin, device='cuda')
    w = torch.randn(cout, cin, k, k, device='cuda') * 0.02
    wl = w.contiguous(memory_format=torch.channels_last)
    w2 = w.reshape(cout, cin).contiguous() if k == 1 else None
    with ops.f32_form(FORM):
        fn = (lambda: ops.conv3x3_f32(x, wl, None, relu=True)) if k == 3 else (lambda: ops.pointwise(x, w2, None, None, True))
        fn(); torch.cuda.synchronize()
        ts = []
        for _ in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
    gf = 2.0 * B * H * W * cin * cout * k * k
    print('  %s: %.1f us  %.1f TF' % ((B, H, W, cin, cout, k), min(ts), gf / min(ts) / 1e6))

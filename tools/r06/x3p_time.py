"""(needs tools/exp/conv_x3_limb_planes.patch applied) times the limb-plane (PRE) 3x3 kernel and the in-loop-split kernel on the RpnHead's P2 level (15 images) and conv4's 3x3;
ODET_LIB_PATH selects a diagnostic build (tools/r06/x3p_diag.sh)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tools._diag
import torch
from tf_eager_object_detection_amd import ops
torch.manual_seed(0)
FORM = sys.argv[1] if len(sys.argv) > 1 else 'x3'
for (B, H, W, cin, cout) in ((15, 200, 334, 256, 512), (15, 50, 84, 256, 256)):
    x = torch.randn(B, H, W, cin, device='cuda')
    wl = (torch.randn(cout, cin, 3, 3, device='cuda') * 0.02).contiguous(memory_format=torch.channels_last)
    with ops.f32_form(FORM):
        ops.split_activation(x)
        out = []
        for fn in (lambda: ops.conv3x3_f32(x, wl, None, relu=True), lambda: ops.conv3x3_f32(x, wl, None, relu=True, inloop=True)):
            fn(); torch.cuda.synchronize()
            ts = []
            for _ in range(4):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); fn(); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            out.append(min(ts))
    gf = 2.0 * B * H * W * cin * cout * 9
    print('  %s: planes %.1f us %.1f TF | in-loop %.1f us %.1f TF' % ((B, H, W, cin, cout), out[0], gf / out[0] / 1e6, out[1], gf / out[1] / 1e6))

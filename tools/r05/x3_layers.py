#!/usr/bin/env python3
"""Split-precision float32 form (csrc/conv_x3.hip) against the exact-float32 form and the float64 truth, layer by layer:
errors (relative to the output's rms) and times on the ResNet-101-FPN shapes of a `--batch`-image pass at 800 x 1333."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from tf_eager_object_detection_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=15)
ap.add_argument('--reps', type=int, default=5)
ap.add_argument('--check', action='store_true', help='errors against float64 on small maps')
ap.add_argument('--out', default='')
ap.add_argument('--forms', default='exact,x3,x2')
args = ap.parse_args()
FORMS = tuple(args.forms.split(','))
torch.manual_seed(0)
dev = 'cuda'


def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(reps):
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return min(ts)


res = {'batch': args.batch, 'layers': []}
if args.check:
    for (B, H, W, cin, cout, k) in [(2, 25, 42, 256, 512, 3), (1, 50, 84, 64, 64, 3), (3, 13, 21, 512, 512, 3), (2, 50, 84, 256, 64, 1),
                                    (2, 25, 42, 1024, 256, 1), (1, 100, 167, 64, 256, 1), (1, 7, 9, 2048, 512, 1), (4, 31, 45, 96, 128, 1)]:
        x = (torch.randn(B, H, W, cin, device=dev) * 3.0).contiguous()
        w = torch.randn(cout, cin, k, k, device=dev) * (2.0 / (cin * k * k)) ** 0.5
        b = torch.randn(cout, device=dev) * 0.1
        wl = w.contiguous(memory_format=torch.channels_last)
        ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), b.double(), padding=k // 2).permute(0, 2, 3, 1)
        outs = {}
        for form in FORMS:
            with ops.f32_form(form):
                outs[form] = (ops.conv3x3_f32(x, wl, b) if k == 3 else ops.pointwise(x, w.reshape(cout, cin).contiguous(), b)).double()
        rms = float(ref.pow(2).mean().sqrt())
        rec = {'shape': [B, H, W, cin, cout, k], 'rms': rms}
        for form in FORMS:
            d = (outs[form] - ref).abs()
            rec[form] = {'max_rel_rms': float(d.max()) / rms, 'mean_rel_rms': float(d.mean()) / rms}
        th = F.conv2d(x.permute(0, 3, 1, 2), w, b, padding=k // 2).permute(0, 2, 3, 1).double()
        d = (th - ref).abs()
        rec['torch_f32'] = {'max_rel_rms': float(d.max()) / rms, 'mean_rel_rms': float(d.mean()) / rms}
        print(json.dumps(rec)); res['layers'].append(rec)
else:
    B = args.batch
    # (name, H, W, cin, cout, k): the distinct shapes of a ResNet-101-FPN pass
    shapes = [('conv2 3x3', 200, 334, 64, 64, 3), ('conv3 3x3', 100, 167, 128, 128, 3), ('conv4 3x3', 50, 84, 256, 256, 3),
              ('conv5 3x3', 25, 42, 512, 512, 3), ('rpn P2', 200, 334, 256, 512, 3), ('smooth P2', 200, 334, 256, 256, 3),
              ('conv2 first 1x1', 200, 334, 256, 64, 1), ('conv2 last 1x1', 200, 334, 64, 256, 1),
              ('conv3 first 1x1', 100, 167, 512, 128, 1), ('conv3 last 1x1', 100, 167, 128, 512, 1),
              ('conv4 first 1x1', 50, 84, 1024, 256, 1), ('conv4 last 1x1', 50, 84, 256, 1024, 1),
              ('conv5 first 1x1', 25, 42, 2048, 512, 1), ('conv5 last 1x1', 25, 42, 512, 2048, 1),
              ('fc1', 1, 1000, 12544, 1024, 1), ('fc2', 1, 1000, 1024, 1024, 1)]
    for name, H, W, cin, cout, k in shapes:
        x = torch.randn(B, H, W, cin, device=dev)
        w = torch.randn(cout, cin, k, k, device=dev) * (2.0 / (cin * k * k)) ** 0.5
        b = torch.randn(cout, device=dev) * 0.1
        wl = w.contiguous(memory_format=torch.channels_last)
        w2 = w.reshape(cout, cin).contiguous() if k == 1 else None
        rec = {'layer': name, 'shape': [B, H, W, cin, cout, k], 'GFLOP': 2.0 * B * H * W * cin * cout * k * k / 1e9}
        for form in FORMS:
            with ops.f32_form(form):
                fn = (lambda: ops.conv3x3_f32(x, wl, b, relu=True)) if k == 3 else (lambda: ops.pointwise(x, w2, b, None, True))
                us = timed(fn, args.reps)
            rec[form + '_us'] = us
            rec[form + '_TFLOPs'] = rec['GFLOP'] / us * 1e-3 * 1e3 / 1e3 * 1e3 / 1e3 if False else rec['GFLOP'] * 1e9 / (us * 1e-6) / 1e12
        print('%-18s ' % name + ' | '.join('%s %8.1f us %6.1f TF' % (f, rec[f + '_us'], rec[f + '_TFLOPs']) for f in FORMS), flush=True)
        res['layers'].append(rec)
if args.out:
    json.dump(res, open(args.out, 'w'), indent=1)

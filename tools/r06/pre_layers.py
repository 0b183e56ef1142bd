#!/usr/bin/env python3
"""(needs tools/exp/conv_x3_limb_planes.patch applied: measured and NOT adopted, round 6)
Round 6: the split-precision 3x3 layers on their input's LIMB PLANES (csrc/conv_x3.hip PRE: both operands by LDS-DMA, no split
in the K loop) against the in-loop split, on the ResNet-101-FPN 3x3 shapes of a `--batch`-image pass; the planes are in place
before the timed launch (they are the producing layer's by-product) and the stand-alone split is timed beside it.
    python tools/r06/pre_layers.py [--batch 15] [--forms x3,x2] [--out file.json]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tf_eager_object_detection_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=15)
ap.add_argument('--reps', type=int, default=5)
ap.add_argument('--forms', default='x3,x2')
ap.add_argument('--out', default='')
args = ap.parse_args()
torch.manual_seed(0)


def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(reps):
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return min(ts)


B = args.batch
shapes = [('conv2 3x3', 200, 334, 64, 64), ('conv3 3x3', 100, 167, 128, 128), ('conv4 3x3', 50, 84, 256, 256),
          ('conv5 3x3', 25, 42, 512, 512), ('rpn P2', 200, 334, 256, 512), ('smooth P2', 200, 334, 256, 256)]
res = {'batch': B, 'layers': []}
for name, H, W, cin, cout in shapes:
    x = torch.randn(B, H, W, cin, device='cuda')
    w = (torch.randn(cout, cin, 3, 3, device='cuda') * (2.0 / (cin * 9)) ** 0.5).contiguous(memory_format=torch.channels_last)
    b = torch.randn(cout, device='cuda') * 0.1
    rec = {'layer': name, 'shape': [B, H, W, cin, cout, 3], 'GFLOP': 2.0 * B * H * W * cin * cout * 9 / 1e9}
    for form in args.forms.split(','):
        with ops.f32_form(form):
            t_in = timed(lambda: ops.conv3x3_f32(x, w, b, relu=True, inloop=True), args.reps)
            ops.limbs_drop(x)
            t_split = timed(lambda: ops.split_activation(x), args.reps)
            assert ops.limbs_of(x) is not None
            t_pre = timed(lambda: ops.conv3x3_f32(x, w, b, relu=True), args.reps)
            t_pre_out = timed(lambda: ops.conv3x3_f32(x, w, b, relu=True, limbs=True), args.reps)
            same = torch.equal(ops.conv3x3_f32(x, w, b, relu=True), ops.conv3x3_f32(x, w, b, relu=True, inloop=True))
            ops.limbs_drop(x)
        tf = lambda us: rec['GFLOP'] * 1e9 / (us * 1e-6) / 1e12
        rec[form] = {'inloop_us': t_in, 'pre_us': t_pre, 'pre_limbs_out_us': t_pre_out, 'split_us': t_split,
                     'inloop_TF': tf(t_in), 'pre_TF': tf(t_pre), 'pre_limbs_out_TF': tf(t_pre_out), 'bit_identical': bool(same)}
        print('%-10s %s  in-loop %8.1f us %6.1f TF | planes %8.1f us %6.1f TF | + planes out %8.1f us %6.1f TF | split alone %7.1f us | same bits %s'
              % (name, form, t_in, tf(t_in), t_pre, tf(t_pre), t_pre_out, tf(t_pre_out), t_split, same), flush=True)
    res['layers'].append(rec)
if args.out:
    json.dump(res, open(args.out, 'w'), indent=1)

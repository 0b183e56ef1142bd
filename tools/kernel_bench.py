#!/usr/bin/env python3
"""Per-entry-point timing at the BASELINE shapes (SURVEY.md 8d protocol: 20 warm-up + 200 timed launches,
HIP events on one stream, median) with the algorithmic bytes of 8(d) -> GB/s.  Latency-bound stages
(NMS, post-ops) are reported in microseconds only.

    python tools/kernel_bench.py [--json out.json]"""
import argparse, json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tf_eager_object_detection_amd import ops, synthetic as syn
from tf_eager_object_detection_amd.pipeline import FpnHotPath, FrcnnHotPath, synthetic_fpn_inputs
from tf_eager_object_detection_amd.utils.anchor_generator import fpn_level_tables, make_fpn_anchors, \
    generate_anchor_base, generate_by_anchor_base_tf


def timeit(fn, warm=20, reps=200):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    return float(np.median(ts))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--json')
    a = ap.parse_args()
    shape = (800, 1333)
    host, dev = synthetic_fpn_inputs(shape, 21, 1000, 256, seed=1234)
    N = syn.num_fpn_anchors(shape)
    rows = []

    def add(name, us, nbytes=None, note=''):
        rows.append(dict(entry=name, us=us, bytes=nbytes, GBps=(nbytes / us / 1e3 if nbytes else None), note=note))

    anchors = make_fpn_anchors(shape, syn.FPN_STRIDES, syn.FPN_BASE_SIZES, syn.FPN_SCALES, syn.FPN_RATIOS)
    add('odet_anchors_fpn (267069 anchors)', timeit(lambda: make_fpn_anchors(shape, syn.FPN_STRIDES, syn.FPN_BASE_SIZES,
                                                                            syn.FPN_SCALES, syn.FPN_RATIOS)), N * 16)
    add('odet_rpn_fg_softmax (FPN layout)', timeit(lambda: ops.rpn_fg_softmax(dev['rpn_logits'], 1, ops.RPN_LAYOUT_FPN)), N * 12)
    fg = ops.rpn_fg_softmax(dev['rpn_logits'], 1, ops.RPN_LAYOUT_FPN)
    out = torch.empty_like(anchors)
    add('odet_decode + clip', timeit(lambda: ops.decode(anchors, dev['rpn_deltas'], [0] * 4, [1] * 4, shape, out=out)), N * 48)
    base = torch.from_numpy(generate_anchor_base().astype(np.float32)).cuda()
    add('odet_anchors_shift (C4 800x1333, 37800)', timeit(lambda: generate_by_anchor_base_tf(base, 16, 50, 84)), 37800 * 16)
    hot = FpnHotPath(shape, 21, 1000, 256)
    add('odet_fpn_proposals (prepare+select+NMS+levels)', timeit(lambda: hot.stage_proposals(dev['rpn_logits'], dev['rpn_deltas'])),
        None, 'latency chain; prepare alone moves %d MB' % (N * 44 // 1000000))
    ws = torch.empty(ops.L.lib().odet_region_proposal_workspace_bytes(N, 1000), dtype=torch.uint8, device='cuda')
    add('odet_region_proposal (anchors + scores given)', timeit(lambda: ops.region_proposal(
        dev['rpn_deltas'], anchors, fg, shape, 1000, 0.7, [0] * 4, [1] * 4, workspace=ws)), None, 'incl. one host sync')
    hot.stage_proposals(dev['rpn_logits'], dev['rpn_deltas'])
    torch.cuda.synchronize()
    k = int(hot.roi_count.item())
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import algorithmic_roi_bytes
    algo = algorithmic_roi_bytes(hot.sorted_rois[:k].cpu().numpy(), hot.roi_level[:k].cpu().numpy(),
                                 syn.fpn_level_shapes(shape)[:4], shape, 256)
    add('odet_roi_order + odet_roi_pool_ordered (FPN, 14x14+max)', timeit(lambda: hot.stage_roi(dev['feats'])), algo['B_roi'],
        'B_roi of SURVEY 8d')
    add('odet_post_ops_record (21 classes)', timeit(lambda: hot.stage_detect(dev['cls_scores'], dev['cls_deltas'])), None,
        'latency chain')
    # single-level variants
    rng = np.random.default_rng(0)
    for name, ch, flag, shp in (('VGG16 600x800 14x14+max C=512', 512, True, (600, 800)),
                                ('ResNet-C4 800x1333 7x7 C=1024', 1024, False, (800, 1333))):
        fr = FrcnnHotPath(shp, 21, 300, ch, max_pooling_flag=flag)
        lg = torch.from_numpy(rng.normal(0, 1.5, (fr.fh * fr.fw, 2 * fr.A)).astype(np.float32)).cuda()
        dl = torch.from_numpy(syn.rpn_deltas(fr.N, rng, 0.1)).cuda()
        ft = torch.from_numpy(rng.standard_normal((1, fr.fh, fr.fw, ch), dtype=np.float32)).cuda()
        add('odet_frcnn_proposals (%s)' % name, timeit(lambda: fr.stage_proposals(lg, dl)), None, 'latency chain')
        outb = 300 * 49 * ch * 4
        add('odet_roi_pool (%s)' % name, timeit(lambda: fr.stage_roi(ft)), outb, 'output bytes only')
    b1 = torch.from_numpy(syn.random_boxes(4096, shape, rng, 16, 400)).cuda()
    b2 = torch.from_numpy(syn.random_boxes(64, shape, rng, 16, 400)).cuda()
    add('odet_pairwise_iou (4096 x 64)', timeit(lambda: ops.pairwise_iou(b1, b2)), 4096 * 64 * 4)
    # dense-path glue kernels at the ResNet-101-FPN 800x1333 shapes, float16 NHWC, batch 4
    B = 4
    g16 = lambda *sh: torch.randn(*sh, device='cuda', dtype=torch.float16)
    x = g16(B, 200, 334, 256); bias = g16(256); res = g16(B, 200, 334, 256)
    add('odet_bias_act (conv2 block output: bias + shortcut + ReLU, fp16, batch 4)', timeit(lambda: ops.bias_act_(x, bias, res, True)),
        3 * x.numel() * 2, 'activation in + shortcut in + out')
    x2 = g16(B, 200, 334, 64); bias2 = g16(64)
    add('odet_bias_act (conv2 1x1 output: bias + ReLU, fp16, batch 4)', timeit(lambda: ops.bias_act_(x2, bias2, None, True)),
        2 * x2.numel() * 2, 'activation in + out')
    top, lat = g16(B, 100, 167, 256), g16(B, 200, 334, 256)
    outm = torch.empty_like(lat)
    add('odet_fpn_topdown_merge (P3 -> P2, fp16, batch 4)', timeit(lambda: ops.fpn_topdown_merge(top, lat, out=outm)),
        (2 * lat.numel() + top.numel()) * 2, 'lateral in + out + coarse map')
    sc = g16(B, 200, 334, 6); so = torch.empty((B, 200 * 334 * 3, 2), dtype=torch.float32, device='cuda'); bs = g16(6)
    add('odet_rpn_pack (P2 scores, fp16 -> fp32, batch 4)', timeit(lambda: ops.rpn_pack(sc, bs, so, 0)), sc.numel() * 6, 'fp16 in + fp32 out')
    # the bottleneck blocks' last 1x1 convolution + bias + shortcut + ReLU on MFMA (batch 8, float16 NHWC)
    for (h_, w_, cin, cout) in ((200, 334, 64, 256), (100, 167, 128, 512), (50, 84, 256, 1024)):
        xx = g16(8, h_, w_, cin); ww = g16(cout, cin) * 0.05; bb = g16(cout); rr = g16(8, h_, w_, cout)
        oo = torch.empty_like(rr)
        add('odet_conv1x1_f16 (%d -> %d at %dx%d + shortcut + ReLU, fp16, batch 8)' % (cin, cout, h_, w_),
            timeit(lambda: ops.conv1x1_f16(xx, ww, bb, rr, True, out=oo), warm=5, reps=50),
            8 * h_ * w_ * (cin + 2 * cout) * 2, 'x + shortcut + y once; %.1f GFLOP' % (2e-9 * 8 * h_ * w_ * cin * cout))
    for r in rows:
        print('%-58s %8.1f us %s %s' % (r['entry'], r['us'], ('%8.1f GB/s' % r['GBps']) if r['GBps'] else ' ' * 13, r['note']))
    if a.json:
        json.dump(rows, open(a.json, 'w'), indent=1)


if __name__ == '__main__':
    main()

#!/usr/bin/env python3
"""Times the RoI kernel alone, in the launch shapes bench.py uses (B-image launch of one stream group, and the
one-image launch), cold (a 1 GiB write went through the caches since the maps were last read) and warm, with HIP
events attached to the dispatch.  Prints one JSON line with the timings and a SHA-256 of the pooled features, so
that two builds of the library (ODET_LIB_PATH) can be compared bit for bit:

    python tools/roi_bench.py                       # in-tree library
    ODET_LIB_PATH=tools/exp/libodet_hip_r01.so python tools/roi_bench.py
"""
import argparse
import hashlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tools._diag            # (ODET_LIB_PATH selects a diagnostic build: tools only, the product reads no environment)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--maps', choices=['f32', 'f16'], default='f32')
    ap.add_argument('--shape', default='800x1333')
    ap.add_argument('--tag', default='')
    ap.add_argument('--flush', choices=['write', 'read'], default='write',
                    help="what goes through the caches before a 'cold' launch: a 1 GiB write (leaves DIRTY lines that the "
                         "launch has to evict) or a 1 GiB read (clean lines)")
    args = ap.parse_args()
    from tf_eager_object_detection_amd import _lib, ops
    from tf_eager_object_detection_amd.pipeline import FpnStepBatch, synthetic_fpn_inputs
    _lib.lib()
    shape = tuple(int(v) for v in args.shape.split('x'))
    B = args.batch
    fdt = torch.float16 if args.maps == 'f16' else torch.float32
    host, dev = synthetic_fpn_inputs(shape, 21, 1000, 256, seed=1234)
    sb = FpnStepBatch(B, shape, 21, 1000, 256, feature_dtype=fdt, blind_chunks=1)
    gen = torch.Generator(device='cuda')
    gen.manual_seed(4321)
    for b in range(B):
        if b == 0:
            d = dict(dev)
            d['feats'] = [f.to(fdt) for f in dev['feats']]
        else:
            pa = torch.randperm(dev['rpn_logits'].shape[0], device='cuda', generator=gen)
            d = dict(rpn_logits=dev['rpn_logits'][pa].contiguous(), rpn_deltas=dev['rpn_deltas'][pa].contiguous(),
                     feats=[torch.randn(f.shape, device='cuda', dtype=torch.float32, generator=gen).to(fdt)
                            for f in dev['feats']],
                     cls_scores=dev['cls_scores'], cls_deltas=dev['cls_deltas'])
        sb.bind(b, d['rpn_logits'], d['rpn_deltas'], d['feats'], d['cls_scores'], d['cls_deltas'])
    sb.enqueue(sb.STAGE_PROPOSALS, B)
    torch.cuda.synchronize()
    assert all(int(h.nms_done.item()) == 1 for h in sb.slots)
    flush = torch.empty(1 << 28, dtype=torch.float32, device='cuda')      # 1 GiB

    def timed(count, cold):
        ts = []
        for _ in range(args.reps):
            if cold:
                if args.flush == 'write':
                    flush.fill_(1.0)
                else:
                    flush.sum()
            a, b_ = ops.ProfEvent(), ops.ProfEvent()
            sb.steps[0].roi_start_event, sb.steps[0].roi_stop_event = a.handle, b_.handle
            sb.enqueue(sb.STAGE_ROI, count)
            torch.cuda.synchronize()
            sb.steps[0].roi_start_event, sb.steps[0].roi_stop_event = None, None
            ts.append(a.elapsed_ms(b_) * 1e3)
        ts = np.array(ts[2:])
        return dict(median_us=float(np.median(ts)), min_us=float(ts.min()), mean_us=float(ts.mean()))

    res = dict(tag=args.tag, flush=args.flush, lib=os.environ.get('ODET_LIB_PATH', 'in-tree'), batch=B, maps=args.maps)
    res['batch_cold'] = timed(B, True)
    res['batch_warm'] = timed(B, False)
    res['one_cold'] = timed(1, True)
    res['one_warm'] = timed(1, False)
    h = hashlib.sha256()
    for b in range(B):
        h.update(sb.roi_features[b].cpu().numpy().tobytes())
    res['features_sha256'] = h.hexdigest()
    res['rois'] = [int(s.roi_count.item()) for s in sb.slots]
    print(json.dumps(res))


if __name__ == '__main__':
    main()

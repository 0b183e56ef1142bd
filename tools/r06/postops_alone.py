#!/usr/bin/env python3
"""round 6: k_postops behind the RoI kernel (the hot path's order: its class workgroups' agent-scope releases find an L2 full of
the RoI features' dirty lines) and alone (stage_detect in a loop), under rocprofv3 --kernel-trace --stats"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tf_eager_object_detection_amd.pipeline import FpnHotPath, synthetic_fpn_inputs
host, dev = synthetic_fpn_inputs((800, 1333), 21, 1000, 256, seed=1234, score_kind='distinct')
hot = FpnHotPath((800, 1333), 21, 1000, 256)
hot.step(dev['rpn_logits'], dev['rpn_deltas'], dev['feats'], dev['cls_scores'], dev['cls_deltas'])
torch.cuda.synchronize()
mode = sys.argv[1]
for _ in range(400):
    if mode == 'alone':
        hot.stage_detect(dev['cls_scores'], dev['cls_deltas'])
    elif mode == 'after_roi':
        hot.stage_roi(dev['feats']); hot.stage_detect(dev['cls_scores'], dev['cls_deltas'])
    else:
        hot.step(dev['rpn_logits'], dev['rpn_deltas'], dev['feats'], dev['cls_scores'], dev['cls_deltas'])
torch.cuda.synchronize()

#!/usr/bin/env python3
"""round 6: the hot path's stages one image at a time (FpnHotPath.step, bench inputs), 300 times, for `rocprofv3 --kernel-trace
--stats`; ODET_LIB_PATH selects a diagnostic build"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tools._diag
import torch
from tf_eager_object_detection_amd.pipeline import FpnHotPath, synthetic_fpn_inputs
host, dev = synthetic_fpn_inputs((800, 1333), 21, 1000, 256, seed=1234, score_kind='distinct')
hot = FpnHotPath((800, 1333), 21, 1000, 256)
for _ in range(400):
    hot.step(dev['rpn_logits'], dev['rpn_deltas'], dev['feats'], dev['cls_scores'], dev['cls_deltas'])
torch.cuda.synchronize()

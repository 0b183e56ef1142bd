#!/usr/bin/env python3
"""round 6: the proposal stage alone (FpnHotPath.stage_proposals, bench inputs), 300 times; run under
`rocprofv3 --kernel-trace --stats` for the per-kernel averages.  ODET_LIB_PATH selects a variant build."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tools._diag
import torch
from tf_eager_object_detection_amd.pipeline import FpnHotPath, synthetic_fpn_inputs
kind = sys.argv[1] if len(sys.argv) > 1 else 'distinct'
host, dev = synthetic_fpn_inputs((800, 1333), 21, 1000, 256, seed=1234, score_kind=kind)
hot = FpnHotPath((800, 1333), 21, 1000, 256, nms_first_chunk=0 if kind == 'distinct' else 2560)
for _ in range(300):
    hot.stage_proposals(dev['rpn_logits'], dev['rpn_deltas'])
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(300):
    hot.stage_proposals(dev['rpn_logits'], dev['rpn_deltas'])
e1.record(); torch.cuda.synchronize()
print(kind, os.environ.get('ODET_LIB_PATH', 'product'), 'proposal stage %.1f us per image' % (e0.elapsed_time(e1) * 1e3 / 300), flush=True)

#!/usr/bin/env python3
"""Diagnostic (needs a build with ODET_EXTRA_HIPCC_FLAGS=-DODET_STAMPS): where k_nms_scan spends its
time on the bench workload.  Stamps are wall_clock64 ticks (100 MHz) written by thread 0 into the
NMS workspace header."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tf_eager_object_detection_amd.pipeline import FpnHotPath, synthetic_fpn_inputs

host, dev = synthetic_fpn_inputs((800, 1333), 21, 1000, 256, seed=1234)
hot = FpnHotPath((800, 1333), 21, 1000, 256)
for _ in range(5):
    hot.stage_proposals(dev['rpn_logits'], dev['rpn_deltas'])
torch.cuda.synchronize()
st = hot.ws_rpn[64:64 + 512].cpu().numpy().view(np.uint64)
names = ['entry', 'prologue loads', 'block loop', 'keptpre+outputs', 'state', 'assign_levels']
for k in range(1, 6):
    print('%-16s %7.2f us' % (names[k], (int(st[k]) - int(st[k - 1])) / 100.0))
print('total %.2f us' % ((int(st[5]) - int(st[0])) / 100.0))
print('  outputs: keptpre+barrier %.2f us, positions %.2f us, stores+levels %.2f us' % ((int(st[6]) - int(st[2])) / 100.0, (int(st[7]) - int(st[6])) / 100.0, (int(st[3]) - int(st[7])) / 100.0))
for w in range(16):
    if st[16 + w]:
        print('wave %2d: own blocks start %7.2f us, end %7.2f us' % (w, (int(st[16 + w]) - int(st[0])) / 100.0,
                                                                    (int(st[32 + w]) - int(st[0])) / 100.0))
dc, dw = int(st[61]) - int(st[60]), int(st[5]) - int(st[0])
print('clock64 delta %d, wall delta %d ticks -> shader clock %.3f GHz (if clock64 = s_memtime)' % (dc, dw, dc / (dw * 10.0)))

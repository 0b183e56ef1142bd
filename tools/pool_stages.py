"""Throughput of the stream pool per stage subset: where does the chip time of an image go?"""
import sys, time, torch, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tf_eager_object_detection_amd.pipeline import FpnStreamPool, synthetic_fpn_inputs
host, dev = synthetic_fpn_inputs((800, 1333), 21, 1000, 256, seed=1234)
S = 8
pool = FpnStreamPool(S, (800, 1333), 21, 1000, 256)
for k in range(S):
    pool.bind(k, dev['rpn_logits'], dev['rpn_deltas'], dev['feats'], dev['cls_scores'], dev['cls_deltas'])
for i in range(4 * S): pool.submit()
pool.wait(); torch.cuda.synchronize()
for name, stages in (('all', 7), ('proposals', 1), ('roi', 2), ('detect', 4), ('proposals+roi', 3), ('roi+detect', 6)):
    N = 1600
    t0 = time.perf_counter()
    for i in range(N): pool.submit(stages=stages)
    pool.wait(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print('%-14s %.1f us/image' % (name, dt / N * 1e6))
pool.close()

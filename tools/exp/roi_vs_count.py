"""RoI-kernel time against the number of RoIs (one image, bench shapes), warm and with flushed caches: tells a
throughput bound (time proportional to the RoIs) from a latency / rounds bound (steps at multiples of the 512
workgroups the chip holds)."""
import sys, torch, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tf_eager_object_detection_amd.pipeline import FpnHotPath, synthetic_fpn_inputs
from tf_eager_object_detection_amd import ops
flush = torch.empty(1 << 28, dtype=torch.float32, device='cuda')
for K in (128, 256, 512, 768, 1000, 1536, 2000):
    host, dev = synthetic_fpn_inputs((800, 1333), 21, K, 256, seed=1234)
    hot = FpnHotPath((800, 1333), 21, K, 256, blind_chunks=3)
    hot.stage_proposals(dev['rpn_logits'], dev['rpn_deltas'])
    torch.cuda.synchronize()
    res = {}
    for mode in ('warm', 'cold'):
        ts = []
        for i in range(20):
            if mode == 'cold':
                flush.add_(1.0)
            e = (ops.ProfEvent(), ops.ProfEvent())
            hot.stage_roi(dev['feats'], events=e)
            torch.cuda.synchronize()
            ts.append(e[0].elapsed_ms(e[1]) * 1e3)
        res[mode] = sorted(ts)[len(ts) // 2]
    print('K=%4d kept=%4d  warm %.1f us  cold %.1f us' % (K, int(hot.roi_count.item()), res['warm'], res['cold']))

import sys, time, torch
sys.path.insert(0, '/root/repo')
from tf_eager_object_detection_amd.pipeline import FpnHotPath, synthetic_fpn_inputs
from tf_eager_object_detection_amd import ops
host, dev = synthetic_fpn_inputs((800, 1333), 21, 1000, 256, seed=1234)
hot = FpnHotPath((800, 1333), 21, 1000, 256)
hot.stage_proposals(dev['rpn_logits'], dev['rpn_deltas'])
ts = []
for i in range(30):
    e = (ops.ProfEvent(), ops.ProfEvent())
    hot.stage_roi(dev['feats'], events=e)
    torch.cuda.synchronize()
    ts.append(e[0].elapsed_ms(e[1]) * 1e3)
print('roi kernel us: min %.1f median %.1f' % (min(ts), sorted(ts)[len(ts) // 2]))

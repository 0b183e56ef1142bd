"""RoI-kernel time of the bench workload, warm (maps resident in the Infinity Cache) and cold (caches
flushed by sweeping a 1 GiB buffer between launches).  Diagnostic builds: `tools/roi_ablate_build.sh`
makes libodet_hip_a{1..4}.so with -DODET_ROI_ABLATE=k (no loads / no lerps / no stores / prologue only);
pass the library path as argv[1] to time one of them."""
import sys, torch, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tf_eager_object_detection_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = sys.argv[1]
from tf_eager_object_detection_amd.pipeline import FpnHotPath, synthetic_fpn_inputs
from tf_eager_object_detection_amd import ops
host, dev = synthetic_fpn_inputs((800, 1333), 21, 1000, 256, seed=1234)
hot = FpnHotPath((800, 1333), 21, 1000, 256)
hot.stage_proposals(dev['rpn_logits'], dev['rpn_deltas'])
flush = torch.empty(1 << 28, dtype=torch.float32, device='cuda')
for mode in ('warm', 'cold'):
    ts = []
    for i in range(30):
        if mode == 'cold':
            flush.add_(1.0)
        e = (ops.ProfEvent(), ops.ProfEvent())
        hot.stage_roi(dev['feats'], events=e)
        torch.cuda.synchronize()
        ts.append(e[0].elapsed_ms(e[1]) * 1e3)
    print('%s %s roi kernel us: min %.1f median %.1f' % (sys.argv[1] if len(sys.argv) > 1 else 'default', mode, min(ts), sorted(ts)[len(ts) // 2]))

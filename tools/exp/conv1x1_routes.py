import sys, torch, collections
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
torch.backends.cudnn.benchmark = True
from tf_eager_object_detection_amd.model import fpn_detector as fd
m = fd.ResNetFpnDetector(101, 21, (800, 1333), 1000, dtype=torch.float16, max_batch=8, blind_chunks=3).prepare()
x = torch.randn(8, 800, 1333, 3, device='cuda')
with torch.no_grad():
    m(x); m(x)
torch.cuda.synchronize()
c = collections.Counter()
for k, v in sorted(fd._GEMM_ROUTE.items(), key=lambda kv: (-kv[0][0][2], kv[0][0][1])):
    print('%4dx%-4d %5d -> %-5d relu=%d shortcut=%d : %s' % (k[0][2], k[0][3], k[0][1], k[1], k[3], k[4], v))

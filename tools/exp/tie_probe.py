import os, sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
os.environ.setdefault('MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD', '0')
from tf_eager_object_detection_amd.model.fpn_detector import ResNetFpnDetector
torch.manual_seed(0)
m = ResNetFpnDetector(101, 21, (800, 1333), 1000, dtype=torch.float16, max_batch=2, blind_chunks=3).prepare()
rng = np.random.default_rng(0)
img = torch.from_numpy((rng.uniform(0, 255, (2, 800, 1333, 3)) - np.float32([103.939, 116.779, 123.68])).astype(np.float32)).cuda()
with torch.no_grad():
    s, d, maps = m._dense(img)
p = torch.softmax(s[0], dim=-1)[:, 1]
v, c = torch.unique(p, return_counts=True)
order = torch.argsort(v, descending=True)
print('anchors', p.numel(), 'distinct scores', v.numel())
cum = 0
for i in order[:12].tolist():
    cum += int(c[i]); print('score %.9f count %d cum %d' % (float(v[i]), int(c[i]), cum))
cs = torch.cumsum(c[order], 0)
for k in (1000, 1536, 4096, 8192, 20000):
    j = int(torch.searchsorted(cs, k))
    print('top-%d reaches distinct value #%d, multiplicity there %d' % (k, j, int(c[order][min(j, len(order)-1)])))
print('max multiplicity', int(c.max()), 'at score', float(v[c.argmax()]))

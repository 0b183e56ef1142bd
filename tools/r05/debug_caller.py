import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tf_eager_object_detection_amd.model.base_fpn_model import ResnetV1Fpn
from tf_eager_object_detection_amd.model.fpn_detector import ResNetFpnDetector
torch.manual_seed(31)
shape, K = (256, 352), 300
m = ResnetV1Fpn(depth=50, rpn_proposal_num_post_nms_test=K, prediction_score_threshold=0.0)
rng = np.random.default_rng(3)
img = torch.from_numpy((rng.uniform(0, 255, (1,) + shape + (3,)) - 110).astype(np.float32)).cuda()
with torch.no_grad():
    c = m._extractor(img); p = m._neck(c)
    print([tuple(x.shape) for x in p], [float(x.float().abs().mean()) for x in p])
    s, d = m._get_fpn_head_results(p)
    print(s.shape, d.shape, float(s.abs().mean()), float(d.abs().mean()))
    a = m._get_anchors(list(shape)); print(a.shape)
    fg = m._fg_scores(s); print(fg.shape, float(fg.min()), float(fg.max()))
    rois = m._rpn_proposal((d, a, fg, list(shape)), training=False); print('rois', rois.shape)
    rl, idx = m._assign_levels(rois); print([r.shape[0] for r in rl])
    f = m._get_roi_features(rl, p, list(shape)); print('feat', f.shape, float(f.abs().mean()))
    sc, bb = m._roi_head(f); print(sc.shape, bb.shape, float(sc.abs().mean()))
    print(m(img, training=False))
det = ResNetFpnDetector(50, 21, shape, K, dtype=torch.float32)
det.load_state_dict(m.dense.state_dict()); det.prepare()
print(det(img)[0][3])

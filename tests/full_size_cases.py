"""Full-size known-answer fixtures ("one full-size fixture hash per config"): the seeded synthetic
inputs of 8(d) at the BASELINE.json shapes, and the SHA-256 of every DISCRETE output of the hot path on them
(kept anchor indices of the RPN NMS, pyramid level of every RoI, class labels of the detections) -- not the
tensors.  tests/golden/full_size_hashes.json holds the digests the CPU oracle produced
(tests/golden/make_full_size_hashes.py); the CPU suite re-derives them from the oracle, the GPU suite from the
HIP path through the C ABI.  Test infrastructure only."""
import hashlib
import json
import os

import numpy as np

from tf_eager_object_detection_amd import synthetic as syn

HASH_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'full_size_hashes.json')
M0, S1, S2 = [0, 0, 0, 0], [1, 1, 1, 1], [0.1, 0.1, 0.2, 0.2]

# name -> the reference configuration it stands for (BASELINE.json configs / SURVEY.md 8d)
CASES = {
    'config1_vgg16_600x800': dict(kind='frcnn', shape=(600, 800), channels=512, max_pool=True, K=300, ncls=21,
                                  per_class=50, per_image=50, seed=11),
    'config2_resnet_c4_800x1333': dict(kind='frcnn', shape=(800, 1333), channels=1024, max_pool=False, K=300, ncls=21,
                                       per_class=50, per_image=50, seed=12),
    'config3_resnet101_fpn_800x1333': dict(kind='fpn', shape=(800, 1333), channels=256, K=1000, ncls=21,
                                           per_class=50, per_image=50, seed=1234),
    'config5_resnet101_fpn_1333x1333_81_classes': dict(kind='fpn', shape=(1333, 1333), channels=256, K=1000, ncls=81,
                                                       per_class=100, per_image=300, seed=15),
}


def digest(a, dtype):
    """SHA-256 of the little-endian bytes of `a` as `dtype` (+ its length, so that () and (0,) differ from nothing)."""
    a = np.ascontiguousarray(np.asarray(a).astype(dtype).astype(np.dtype(dtype).newbyteorder('<')))
    return hashlib.sha256(str(a.shape[0]).encode() + b':' + a.tobytes()).hexdigest()


def make_inputs(name):
    """numpy inputs of one case (no feature maps: none of the hashed outputs depends on them)."""
    c = CASES[name]
    rng = np.random.default_rng(c['seed'])
    shape = c['shape']
    if c['kind'] == 'fpn':
        n = syn.num_fpn_anchors(shape)
        deltas = syn.rpn_deltas(n, rng, 0.1)
        logits = syn.logits_from_prob(syn.scores_distinct(n, rng), rng)
    else:
        fh, fw = -(-shape[0] // 16), -(-shape[1] // 16)
        n = fh * fw * 9
        logits = rng.normal(0, 1.5, (fh * fw, 18)).astype(np.float32)
        deltas = syn.rpn_deltas(n, rng, 0.1)
    return dict(rpn_logits=logits, rpn_deltas=deltas, cls_scores=syn.class_scores(c['K'], c['ncls'], rng),
                cls_deltas=syn.class_deltas(c['K'], c['ncls'], rng), num_anchors=n)


def oracle_outputs(name, inp):
    """The discrete outputs, from the CPU oracle (C restatement) stage by stage."""
    from oracle import c_oracle as co
    from oracle import oracle_np as on
    c = CASES[name]
    shape = c['shape']
    if c['kind'] == 'fpn':
        anchors = co.fpn_anchors(shape)
        fg = co.rpn_fg_fpn(inp['rpn_logits'])
    else:
        fh, fw = -(-shape[0] // 16), -(-shape[1] // 16)
        base = on.generate_anchor_base(16, (0.5, 1, 2), (8, 16, 32)).astype(np.float32)
        anchors = on.generate_by_anchor_base_tf(base, 16, fh, fw)
        fg = on.rpn_fg_scores_frcnn(inp['rpn_logits'], 9)
    rois, kept = co.region_proposal(inp['rpn_deltas'], anchors, fg, shape, c['K'], 0.7)
    k = len(kept)
    out = dict(kept_anchor_idx=kept)
    if c['kind'] == 'fpn':
        lvl, perm, _ = co.assign_levels(rois)
        out['roi_level'] = np.asarray(lvl)[perm]     # level (2..5) of every RoI, in the level-sorted order
        out['level_perm'] = np.asarray(perm)
        rois_in = rois[perm]                      # base_fpn_model.py:232-260: the RoI head sees the level-sorted RoIs
    else:
        rois_in = rois
    _, labels, _ = co.post_ops(inp['cls_scores'][:k], inp['cls_deltas'][:k], rois_in, shape, M0, S2, c['per_class'],
                               c['per_image'], 0.3, 0.0, 16, c['ncls'])
    out['det_labels'] = np.zeros(0, np.int32) if labels is None else labels
    return out


DTYPES = dict(kept_anchor_idx=np.int64, roi_level=np.int32, level_perm=np.int64, det_labels=np.int32)


def digests(outputs):
    d = {k: digest(v, DTYPES[k]) for k, v in outputs.items()}
    d['counts'] = {k: int(np.asarray(v).shape[0]) for k, v in outputs.items()}
    return d


def load_golden():
    with open(HASH_FILE) as f:
        return json.load(f)

"""Synthetic inputs for the detection hot path (numpy, host side) -- SURVEY.md section 8(d).

Used by bench.py and the tests; there are no datasets or checkpoints offline, so the hot path
is driven with seeded random tensors of the shapes the reference's models produce.
"""
import math

import numpy as np

FPN_STRIDES = (4, 8, 16, 32, 64)
FPN_BASE_SIZES = (32, 64, 128, 256, 512)
FPN_RATIOS = (0.5, 1.0, 2.0)
FPN_SCALES = (1.,)


def fpn_level_shapes(image_shape, strides=FPN_STRIDES):
    """ceil(H/stride) x ceil(W/stride) per level (reference base_fpn_model.py:178-179)."""
    return [(int(math.ceil(image_shape[0] / s)), int(math.ceil(image_shape[1] / s))) for s in strides]


def num_fpn_anchors(image_shape, strides=FPN_STRIDES, anchors_per_cell=3):
    return sum(h * w for h, w in fpn_level_shapes(image_shape, strides)) * anchors_per_cell


def features(shapes, channels, rng):
    """NHWC float32 normal(0,1) feature maps [1,H,W,C]."""
    return [rng.standard_normal((1, h, w, channels), dtype=np.float32) for h, w in shapes]


def rpn_deltas(n, rng, sigma=0.1):
    """trained-like RPN regression outputs: normal(0, sigma) clipped to +-1."""
    return np.clip(rng.normal(0.0, sigma, size=(n, 4)), -1.0, 1.0).astype(np.float32)


def scores_distinct(n, rng):
    """all-distinct scores in (0,1): (perm + 0.5) / n  (no ties -> tie rule irrelevant)."""
    return ((rng.permutation(n).astype(np.float64) + 0.5) / n).astype(np.float32)


def scores_tied(n, rng, decimals=3):
    return np.round(scores_distinct(n, rng), decimals).astype(np.float32)


def _iou_max(anchors, objs):
    a = anchors.astype(np.float32)
    best = np.zeros(a.shape[0], np.float32)
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    for o in objs:
        iw = np.maximum(0, np.minimum(a[:, 2], o[2]) - np.maximum(a[:, 0], o[0]))
        ih = np.maximum(0, np.minimum(a[:, 3], o[3]) - np.maximum(a[:, 1], o[1]))
        inter = iw * ih
        iou = inter / np.maximum(area_a + (o[2] - o[0]) * (o[3] - o[1]) - inter, 1e-6)
        best = np.maximum(best, iou.astype(np.float32))
    return best


def random_boxes(n, image_shape, rng, min_size=16.0, max_size=800.0):
    """centres uniform, sqrt(area) log-uniform in [min,max], aspect log-uniform in [0.5,2], clipped."""
    H, W = image_shape
    cx = rng.uniform(0, W, n)
    cy = rng.uniform(0, H, n)
    s = np.exp(rng.uniform(np.log(min_size), np.log(max_size), n))
    ar = np.exp(rng.uniform(np.log(0.5), np.log(2.0), n))
    w = s * np.sqrt(ar)
    h = s / np.sqrt(ar)
    b = np.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], axis=1)
    b[:, 0::2] = np.clip(b[:, 0::2], 0, W - 1)
    b[:, 1::2] = np.clip(b[:, 1::2], 0, H - 1)
    return b.astype(np.float32)


def scores_clustered(anchors, image_shape, rng, num_objects=12):
    """trained-like objectness: high for anchors overlapping a few random 'objects', so that the
    best-scoring anchors form dense clusters and greedy NMS has real suppression work."""
    objs = random_boxes(num_objects, image_shape, rng, min_size=48.0, max_size=min(image_shape) * 0.8)
    iou = _iou_max(anchors, objs)
    noise = rng.uniform(0.0, 1.0, anchors.shape[0]).astype(np.float32)
    s = 0.02 + 0.9 * np.sqrt(iou) + 0.05 * noise
    return np.clip(s, 1e-6, 1.0 - 1e-6).astype(np.float32)


def logits_from_prob(p, rng=None):
    """(bg, fg) logits whose softmax fg-probability is ~p (bg logit random, fg = bg + logit(p))."""
    p = np.clip(p.astype(np.float64), 1e-7, 1 - 1e-7)
    bg = np.zeros_like(p) if rng is None else rng.normal(0, 1, p.shape)
    fg = bg + np.log(p / (1 - p))
    return np.stack([bg, fg], axis=1).astype(np.float32)


def class_scores(r, num_classes, rng, sigma=2.0):
    x = rng.normal(0.0, sigma, size=(r, num_classes))
    x = x - x.max(axis=1, keepdims=True)
    e = np.exp(x)
    return (e / e.sum(axis=1, keepdims=True)).astype(np.float32)


def class_deltas(r, num_classes, rng):
    return rng.normal(0.0, 1.0, size=(r, num_classes, 4)).astype(np.float32)

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


def eval_image(rng, raw_shape=(375, 500), min_edge=600, max_edge=1000, num_rois=300, num_classes=21,
               roi_stds=(0.1, 0.1, 0.2, 0.2)):
    """One synthetic evaluation image: ground truth in RAW-image pixels plus what a (noisy but
    sensible) detector's im_detect would return for it in RESIZED-image pixels -- RoIs that are
    jittered ground-truth boxes or clutter, class scores peaked on the right label for the good ones,
    deltas that point (noisily) from the RoI to its object.  Resize rule of the reference's input
    pipeline: shorter side -> min_edge, capped by max_edge on the longer side
    (dataset/utils/tf_dataset_utils.py:129-154).
    -> dict(raw_h, raw_w, img_scale, gt_boxes [g,4], gt_labels [g], scores [R,C], deltas [R,C*4], rois [R,4])"""
    raw_h, raw_w = raw_shape
    img_scale = np.float32(min(min_edge / min(raw_h, raw_w), max_edge / max(raw_h, raw_w)))
    g = int(rng.integers(1, 7))
    cx = rng.uniform(0.15, 0.85, g) * raw_w
    cy = rng.uniform(0.15, 0.85, g) * raw_h
    w = rng.uniform(0.1, 0.5, g) * raw_w
    h = rng.uniform(0.1, 0.5, g) * raw_h
    gt = np.stack([np.clip(cx - w / 2, 0, raw_w - 1), np.clip(cy - h / 2, 0, raw_h - 1),
                   np.clip(cx + w / 2, 0, raw_w - 1), np.clip(cy + h / 2, 0, raw_h - 1)], axis=1).astype(np.float32)
    labels = rng.integers(1, num_classes, g).astype(np.int32)
    owner = rng.integers(-1, g, num_rois)                      # -1 = clutter
    rois = np.empty((num_rois, 4), np.float32)
    logits = rng.normal(0, 1, (num_rois, num_classes)).astype(np.float32)
    deltas = (rng.normal(0, 0.3, (num_rois, num_classes, 4))).astype(np.float32)
    stds = np.asarray(roi_stds, np.float32)
    for r in range(num_rois):
        if owner[r] < 0:
            bw, bh = rng.uniform(20, 0.6 * raw_w), rng.uniform(20, 0.6 * raw_h)
            x1, y1 = rng.uniform(0, raw_w - bw), rng.uniform(0, raw_h - bh)
            box = np.float32([x1, y1, x1 + bw, y1 + bh])
            logits[r, 0] += 3.0
        else:
            b = gt[owner[r]]
            bw, bh = b[2] - b[0] + 1, b[3] - b[1] + 1
            jit = rng.normal(0, 0.12, 4) * np.float32([bw, bh, bw, bh])
            box = (b + jit).astype(np.float32)
            box = np.float32([min(box[0], box[2] - 4), min(box[1], box[3] - 4), box[2], box[3]])
            lab = labels[owner[r]]
            logits[r, lab] += rng.uniform(2.0, 6.0)
            # deltas that lead from the RoI to the object (bbox_transform.py:4-29), plus noise
            rw, rh = box[2] - box[0] + 1, box[3] - box[1] + 1
            rcx, rcy = box[0] + 0.5 * rw, box[1] + 0.5 * rh
            gcx, gcy = b[0] + 0.5 * bw, b[1] + 0.5 * bh
            t = np.float32([(gcx - rcx) / rw, (gcy - rcy) / rh, np.log(bw / rw), np.log(bh / rh)]) / stds
            deltas[r, lab] = t + rng.normal(0, 0.15, 4).astype(np.float32)
        rois[r] = box * img_scale                               # im_detect sees resized-image pixels
    e = np.exp(logits - logits.max(axis=1, keepdims=True))
    scores = (e / e.sum(axis=1, keepdims=True)).astype(np.float32)
    return dict(raw_h=float(raw_h), raw_w=float(raw_w), img_scale=float(img_scale), gt_boxes=gt, gt_labels=labels,
                scores=scores, deltas=deltas.reshape(num_rois, num_classes * 4).astype(np.float32), rois=rois)

"""ctypes front-end of the C oracle (oracle/oracle.c).  TEST INFRASTRUCTURE ONLY.

numpy in / numpy out; every wrapper mirrors one reference function (see oracle.c for the
file:line citations).  Used for full-size parity checks and as bench.py's cpu_baseline."""
import ctypes as C

import numpy as np

from . import build as _build

_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(_build.build())
        _lib.orc_nms.restype = C.c_int
        _lib.orc_clip_filter.restype = C.c_int
        _lib.orc_range_filter.restype = C.c_int
        _lib.orc_region_proposal.restype = C.c_int
        _lib.orc_post_ops.restype = C.c_int
        _lib.orc_max_threads.restype = C.c_int
    return _lib


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def max_threads():
    return int(lib().orc_max_threads())


def anchors_shift(base, stride, fh, fw):
    base = _f(base).reshape(-1, 4)
    A = base.shape[0]
    out = np.empty((fh * fw * A, 4), np.float32)
    lib().orc_anchors_shift(_p(base), C.c_int(A), C.c_int(stride), C.c_int(fh), C.c_int(fw), _p(out))
    return out


def make_anchors(base_size, scales, ratios, fh, fw, stride):
    scales = _f(scales).reshape(-1)
    ratios = _f(ratios).reshape(-1)
    fh, fw = int(np.ceil(fh)), int(np.ceil(fw))
    out = np.empty((fh * fw * scales.size * ratios.size, 4), np.float32)
    lib().orc_make_anchors(C.c_float(base_size), _p(scales), C.c_int(scales.size), _p(ratios),
                           C.c_int(ratios.size), C.c_int(fh), C.c_int(fw), C.c_int(stride), _p(out))
    return out


def fpn_anchors(image_shape, strides=(4, 8, 16, 32, 64), base_sizes=(32, 64, 128, 256, 512),
                ratios=(0.5, 1.0, 2.0), scales=(1.,)):
    return np.concatenate([make_anchors(b, scales, ratios, np.ceil(image_shape[0] / s),
                                        np.ceil(image_shape[1] / s), s)
                           for s, b in zip(strides, base_sizes)], axis=0)


def decode(anchors, deltas, means, stds):
    anchors = _f(anchors).reshape(-1, 4)
    deltas = _f(deltas).reshape(-1, 4)
    out = np.empty_like(anchors)
    lib().orc_decode(_p(anchors), _p(deltas), C.c_int(anchors.shape[0]), _p(_f(means)), _p(_f(stds)), _p(out))
    return out


def encode(src, dst, means, stds):
    src = _f(src).reshape(-1, 4)
    dst = _f(dst).reshape(-1, 4)
    out = np.empty_like(src)
    lib().orc_encode(_p(src), _p(dst), C.c_int(src.shape[0]), _p(_f(means)), _p(_f(stds)), _p(out))
    return out


def clip_filter(boxes, min_value, max_h, max_w, min_edge=None):
    boxes = _f(boxes).reshape(-1, 4)
    n = boxes.shape[0]
    out = np.empty_like(boxes)
    idx = np.empty(n, np.int64)
    m = lib().orc_clip_filter(_p(boxes), C.c_int(n), C.c_float(min_value), C.c_int(max_h), C.c_int(max_w),
                              C.c_float(-1.0 if min_edge is None else min_edge), _p(out), _p(idx))
    return out[:m].copy(), idx[:m].copy()


def range_filter(anchors, max_h, max_w):
    anchors = _f(anchors).reshape(-1, 4)
    idx = np.empty(anchors.shape[0], np.int64)
    m = lib().orc_range_filter(_p(anchors), C.c_int(anchors.shape[0]), C.c_int(max_h), C.c_int(max_w), _p(idx))
    return idx[:m].copy()


def pairwise_iou(b1, b2):
    b1 = _f(b1).reshape(-1, 4)
    b2 = _f(b2).reshape(-1, 4)
    out = np.empty((b1.shape[0], b2.shape[0]), np.float32)
    lib().orc_pairwise_iou(_p(b1), C.c_int(b1.shape[0]), _p(b2), C.c_int(b2.shape[0]), _p(out))
    return out


def nms(boxes, scores, max_out, thr, return_stats=False):
    boxes = _f(boxes).reshape(-1, 4)
    scores = _f(scores).reshape(-1)
    n = boxes.shape[0]
    out = np.empty(max(min(max_out, n), 1), np.int32)
    stats = np.zeros(2, np.int64)
    k = lib().orc_nms(_p(boxes), _p(scores), C.c_int(n), C.c_int(max_out), C.c_float(thr), _p(out), _p(stats))
    return (out[:k].copy(), stats) if return_stats else out[:k].copy()


def crop_and_resize(image, boxes, crop, threads=1):
    image = _f(image)
    if image.ndim == 4:
        image = image[0]
    H, W, Cc = image.shape
    boxes = _f(boxes).reshape(-1, 4)
    R = boxes.shape[0]
    out = np.empty((R, crop[0], crop[1], Cc), np.float32)
    lib().orc_crop_and_resize(_p(image), C.c_int(H), C.c_int(W), C.c_int(Cc), _p(boxes), C.c_int(R),
                              C.c_int(crop[0]), C.c_int(crop[1]), _p(out), C.c_int(threads))
    return out


def roi_pool(feat, rois, stride=None, image_shape=None, pool=7, max_pool=True, threads=1, scratch=None):
    """stride variant (RoiPoolingCropAndResize) when image_shape is None, FPN variant
    (RoiPoolingCropAndResize2) otherwise."""
    feat = _f(feat)
    if feat.ndim == 4:
        feat = feat[0]
    H, W, Cc = feat.shape
    rois = _f(rois).reshape(-1, 4)
    R = rois.shape[0]
    out = np.empty((R, pool, pool, Cc), np.float32)
    if max_pool and scratch is None:
        scratch = np.empty((R, 2 * pool, 2 * pool, Cc), np.float32)
    ih, iw = (0, 0) if image_shape is None else (int(image_shape[0]), int(image_shape[1]))
    lib().orc_roi_pool(_p(feat), C.c_int(H), C.c_int(W), C.c_int(Cc), _p(rois), C.c_int(R),
                       C.c_float(1.0 if stride is None else stride), C.c_int(ih), C.c_int(iw),
                       C.c_int(pool), C.c_int(1 if max_pool else 0), _p(out),
                       _p(scratch) if scratch is not None else None, C.c_int(threads))
    return out


def roi_align(feat, rois, stride, pool=7, threads=1):
    feat = _f(feat)
    if feat.ndim == 4:
        feat = feat[0]
    H, W, Cc = feat.shape
    rois = _f(rois).reshape(-1, 4)
    R = rois.shape[0]
    out = np.empty((R, pool, pool, Cc), np.float32)
    scratch = np.empty((R, 2 * pool, 2 * pool, Cc), np.float32)
    padded = np.empty((H + 2, W + 2, Cc), np.float32)
    lib().orc_roi_align(_p(feat), C.c_int(H), C.c_int(W), C.c_int(Cc), _p(rois), C.c_int(R),
                        C.c_float(stride), C.c_int(pool), _p(out), _p(scratch), _p(padded), C.c_int(threads))
    return out


def softmax(logits):
    logits = _f(logits)
    out = np.empty_like(logits)
    lib().orc_softmax(_p(logits), C.c_int(int(np.prod(logits.shape[:-1]))), C.c_int(logits.shape[-1]), _p(out))
    return out


def rpn_fg_fpn(logits):
    logits = _f(logits).reshape(-1, 2)
    out = np.empty(logits.shape[0], np.float32)
    lib().orc_rpn_fg_fpn(_p(logits), C.c_int(logits.shape[0]), _p(out))
    return out


def rpn_fg_frcnn(logits, A):
    logits = _f(logits).reshape(-1, 2 * A)
    out = np.empty(logits.shape[0] * A, np.float32)
    lib().orc_rpn_fg_frcnn(_p(logits), C.c_int(logits.shape[0]), C.c_int(A), _p(out))
    return out


def assign_levels(rois, min_level=2, max_level=5):
    rois = _f(rois).reshape(-1, 4)
    R = rois.shape[0]
    levels = np.empty(R, np.int32)
    perm = np.empty(R, np.int64)
    counts = np.zeros(max_level - min_level + 1, np.int32)
    lib().orc_assign_levels(_p(rois), C.c_int(R), C.c_int(min_level), C.c_int(max_level),
                            _p(levels), _p(perm), _p(counts))
    return levels, perm, counts


def region_proposal(deltas, anchors, scores, image_shape, K, thr=0.7, means=(0, 0, 0, 0),
                    stds=(1, 1, 1, 1), return_stats=False):
    deltas = _f(deltas).reshape(-1, 4)
    anchors = _f(anchors).reshape(-1, 4)
    scores = _f(scores).reshape(-1)
    n = anchors.shape[0]
    scratch = np.empty((n, 4), np.float32)
    rois = np.empty((max(K, 1), 4), np.float32)
    idx = np.empty(max(K, 1), np.int32)
    stats = np.zeros(2, np.int64)
    k = lib().orc_region_proposal(_p(deltas), _p(anchors), _p(scores), C.c_int(n), C.c_int(image_shape[0]),
                                  C.c_int(image_shape[1]), _p(_f(means)), _p(_f(stds)), C.c_int(K),
                                  C.c_float(thr), _p(scratch), _p(rois), _p(idx), _p(stats))
    res = (rois[:k].copy(), idx[:k].copy())
    return res + (stats,) if return_stats else res


def post_ops(S, D, rois, image_shape, means, stds, max_per_class=50, max_per_image=150, nms_thr=0.3,
             score_thr=0.05, extractor_stride=16, num_classes=21):
    S = _f(S)
    R, Ccls = S.shape
    D = _f(D).reshape(R, Ccls, 4)
    rois = _f(rois).reshape(-1, 4)
    if means is None:
        means = [0, 0, 0, 0]
    if stds is None:
        stds = [1, 1, 1, 1]
    ob = np.empty((max(max_per_image, 1), 4), np.float32)
    oc = np.empty(max(max_per_image, 1), np.int32)
    os_ = np.empty(max(max_per_image, 1), np.float32)
    m = lib().orc_post_ops(_p(S), _p(D), _p(rois), C.c_int(R), C.c_int(Ccls), C.c_int(image_shape[0]),
                           C.c_int(image_shape[1]), _p(_f(means)), _p(_f(stds)), C.c_int(max_per_class),
                           C.c_int(max_per_image), C.c_float(nms_thr), C.c_float(score_thr),
                           C.c_float(extractor_stride), C.c_int(num_classes), _p(ob), _p(oc), _p(os_))
    if m == 0:
        return None, None, None
    return ob[:m].copy(), oc[:m].copy(), os_[:m].copy()

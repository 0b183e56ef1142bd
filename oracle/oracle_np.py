"""CPU ORACLE (numpy restatement) -- TEST INFRASTRUCTURE ONLY, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module, and only as the checker.  The product package
(``tf_eager_object_detection_amd``) never imports anything under ``oracle/``.

What it is
----------
A line-by-line numpy restatement of the Faster-R-CNN / FPN inference hot path of
irvingzhang0512/tf_eager_object_detection (region_proposal.py -> roi_pooling.py ->
prediction.py plus anchor_generator / bbox_tf / bbox_transform) together with the
TensorFlow-1.x CPU kernel semantics those files delegate to
(NonMaxSuppressionV3, CropAndResize, TopKV2, Softmax, MaxPool/AvgPool 2x2, MirrorPad).
Every function cites the reference ``file:line`` it follows (paths relative to the
reference checkout).

PARITY PIN STATUS
-----------------
* PINNED against the reference's own code (executed in the build container, vectors in
  ``tests/golden/ref_numpy_vectors.npz``, generator ``tests/golden/make_ref_vectors.py``):
  ``generate_anchor_base`` (+helpers), ``generate_by_anchor_base_np``, the +1 IoU formula
  (against ``utils/bbox_np.py``), ``voc_ap`` and ``voc_eval_arrays`` (against the reference's ``voc_eval`` run on a
  synthetic dataset in the VOC devkit's file formats: rec / prec / AP of both metrics, bit for bit).
* PARITY UNPINNED for everything that the reference delegates to TensorFlow ops
  (NMS, crop_and_resize, top_k, softmax, exp/log): TensorFlow is not installable in the
  build container and the reference has no tests / fixtures.  Those parts restate the
  published TF r1.13 CPU kernels (third-party dependency, un-vendored, un-pinned by the
  reference; version bracket TF 1.12-1.14) and are anchored on
  the reference's call sites.

Float discipline
----------------
All arithmetic is float32, one IEEE operation per numpy call, in the reference's operation
order (no algebraic simplification, no FMA).  ``exp`` / ``log`` are defined as the
*correctly rounded float32* function (computed through float64); Eigen's packet exp/log
differ from that by <= 1-2 ulp.  Ties in NMS / top-k are broken (score desc, index asc),
which is TF >= 2.2's documented order (r1.13's heap order is unspecified for ties).
"""
import heapq

import numpy as np

F32 = np.float32


# --------------------------------------------------------------------------------------
# elementary float32 helpers
# --------------------------------------------------------------------------------------
def f32(x):
    return np.asarray(x, dtype=np.float32)


def exp32(x):
    """Correctly rounded float32 exp (TF: Eigen pexp, <= ~1 ulp from this)."""
    return np.exp(f32(x).astype(np.float64)).astype(np.float32)


def log32(x):
    """Correctly rounded float32 log."""
    with np.errstate(divide='ignore', invalid='ignore'):
        return np.log(f32(x).astype(np.float64)).astype(np.float32)


# --------------------------------------------------------------------------------------
# utils/anchor_generator.py
# --------------------------------------------------------------------------------------
def generate_anchor_base(base_size=16, ratios=(0.5, 1, 2), scales=2 ** np.arange(3, 6)):
    """utils/anchor_generator.py:63-81 with helpers :84-134 (py-faster-rcnn base anchors),
    restated in closed form.  Reference window is (0,0,b-1,b-1): w = h = b, centre
    c = (b-1)/2.  Per ratio r: ws = round(sqrt(b*b/r)), hs = round(ws*r) (float64,
    numpy round-half-even, :118-120).  The ratio anchor is re-measured by _whctrs (:84-93),
    which returns exactly (ws, hs, c, c) because every intermediate is a multiple of 0.5,
    then per scale s the box is c -/+ 0.5*(ws*s - 1), c -/+ 0.5*(hs*s - 1) (:125-134,
    :96-108).  Row order: ratio-major, scale-minor (:78-80).  Returns float64 [A,4]."""
    r = np.asarray(ratios, dtype=np.float64)
    sc = np.asarray(scales, dtype=np.float64)
    b = float(base_size)
    c = 0.5 * (b - 1.0)
    ws = np.round(np.sqrt((b * b) / r))
    hs = np.round(ws * r)
    W = (ws[:, None] * sc[None, :]).reshape(-1)
    Hh = (hs[:, None] * sc[None, :]).reshape(-1)
    return np.stack([c - 0.5 * (W - 1), c - 0.5 * (Hh - 1),
                     c + 0.5 * (W - 1), c + 0.5 * (Hh - 1)], axis=1)


def generate_by_anchor_base_np(anchor_base, feat_stride, height, width):
    """utils/anchor_generator.py:23-43 -- literal behaviour, including its quirks: height /
    width are PIXELS (arange(0, size, stride), :32-33), and the shift added to the
    (x,y,x,y) base is (y,x,y,x) (:35-36).  Cells row-major (y outer), anchor-minor."""
    ys = np.arange(0, height, feat_stride)
    xs = np.arange(0, width, feat_stride)
    gy = np.repeat(ys, xs.size)
    gx = np.tile(xs, ys.size)
    shift = np.stack([gy, gx, gy, gx], axis=1)                   # [K,4]
    out = shift[:, None, :] + np.asarray(anchor_base)[None, :, :]
    return out.reshape(-1, 4).astype(np.float32)


def generate_by_anchor_base_tf(anchor_base, feat_stride, height, width):
    """utils/anchor_generator.py:46-60.  int32 shifts, row-major (y outer, x inner),
    anchor[k*A+a] = base[a] + (sx,sy,sx,sy)[k]; float32 add."""
    height, width = int(height), int(width)
    sx = (np.arange(width, dtype=np.int32) * np.int32(feat_stride))
    sy = (np.arange(height, dtype=np.int32) * np.int32(feat_stride))
    sx, sy = np.meshgrid(sx, sy)               # tf.meshgrid default indexing='xy'
    sx = sx.reshape(-1)
    sy = sy.reshape(-1)
    shifts = np.stack([sx, sy, sx, sy], axis=1).astype(np.float32)   # [K,4]
    base = f32(anchor_base).reshape(1, -1, 4)
    return (base + shifts[:, None, :]).reshape(-1, 4).astype(np.float32)


def make_anchors(base_anchor_size, anchor_scales, anchor_ratios,
                 featuremap_height, featuremap_width, stride):
    """utils/anchor_generator.py:137-178 (make_anchors + enum_scales + enum_ratios).

    Note the swap: enum_ratios returns (hs, ws) (:178) and the caller receives them as
    (ws, hs) (:143), so w = S*s*sqrt(r), h = S*s/sqrt(r).  tf.meshgrid flattens its
    arguments, which makes the order location-major (row-major y,x) / anchor-minor.
    """
    base_anchor = f32([0, 0, base_anchor_size, base_anchor_size])
    scales = f32(anchor_scales).reshape(-1, 1)
    scaled = base_anchor[None, :] * scales                      # enum_scales :165-167
    ws_in = scaled[:, 2]
    hs_in = scaled[:, 3]
    sqrt_ratios = np.sqrt(f32(anchor_ratios))                   # :173
    ws_r = (ws_in[None, :] / sqrt_ratios[:, None]).reshape(-1)  # :175
    hs_r = (hs_in[None, :] * sqrt_ratios[:, None]).reshape(-1)  # :176
    ws, hs = hs_r, ws_r                                         # :178 + :143 (swap)

    fw = int(np.ceil(float(featuremap_width)))
    fh = int(np.ceil(float(featuremap_height)))
    x_centers = np.arange(fw, dtype=np.float32) * F32(stride)   # :146
    y_centers = np.arange(fh, dtype=np.float32) * F32(stride)   # :147
    x_centers, y_centers = np.meshgrid(x_centers, y_centers)    # :149
    x_centers = x_centers.reshape(-1)
    y_centers = y_centers.reshape(-1)
    ws_g, xc_g = np.meshgrid(ws, x_centers)                     # :151  -> [K, A]
    hs_g, yc_g = np.meshgrid(hs, y_centers)                     # :152
    centers = np.stack([xc_g, yc_g], axis=2).reshape(-1, 2)     # :154-155
    sizes = np.stack([ws_g, hs_g], axis=2).reshape(-1, 2)       # :157-158
    half = F32(0.5) * sizes
    return np.concatenate([centers - half, centers + half], axis=1).astype(np.float32)  # :160-161


def fpn_anchors(image_shape, strides=(4, 8, 16, 32, 64), base_sizes=(32, 64, 128, 256, 512),
                ratios=(0.5, 1.0, 2.0), scales=(1.,)):
    """model/fpn/base_fpn_model.py:163-186 (_get_anchors): per-level make_anchors with
    ceil(H/stride) x ceil(W/stride) cells, concatenated P2->P6."""
    out = []
    for s, b in zip(strides, base_sizes):
        fh = np.ceil(image_shape[0] / s)
        fw = np.ceil(image_shape[1] / s)
        out.append(make_anchors(b, scales, ratios, fh, fw, s))
    return np.concatenate(out, axis=0)


# --------------------------------------------------------------------------------------
# utils/bbox_transform.py
# --------------------------------------------------------------------------------------
def encode_bbox_with_mean_and_std(src_bbox, dst_bbox, target_means, target_stds):
    """utils/bbox_transform.py:4-29."""
    means = f32(target_means)
    stds = f32(target_stds)
    box = f32(src_bbox)
    gt = f32(dst_bbox)
    width = box[..., 2] - box[..., 0] + F32(1.0)
    height = box[..., 3] - box[..., 1] + F32(1.0)
    cx = box[..., 0] + F32(0.5) * width
    cy = box[..., 1] + F32(0.5) * height
    gw = gt[..., 2] - gt[..., 0] + F32(1.0)
    gh = gt[..., 3] - gt[..., 1] + F32(1.0)
    gcx = gt[..., 0] + F32(0.5) * gw
    gcy = gt[..., 1] + F32(0.5) * gh
    dx = (gcx - cx) / width
    dy = (gcy - cy) / height
    dw = log32(gw / width)
    dh = log32(gh / height)
    delta = np.stack([dx, dy, dw, dh], axis=-1)
    return ((delta - means) / stds).astype(np.float32)


def decode_bbox_with_mean_and_std(anchors, bboxes_txtytwth, target_means, target_stds):
    """utils/bbox_transform.py:32-55.  NB x2 = x1 + w (no -1), no clamp on dw/dh."""
    means = f32(target_means)
    stds = f32(target_stds)
    anchors = f32(anchors).reshape(-1, 4)
    delta = f32(bboxes_txtytwth).reshape(-1, 4) * stds + means          # :37
    width = anchors[:, 2] - anchors[:, 0] + F32(1)                      # :40
    height = anchors[:, 3] - anchors[:, 1] + F32(1)                     # :41
    cx = anchors[:, 0] + F32(0.5) * width                               # :42
    cy = anchors[:, 1] + F32(0.5) * height                              # :43
    cx = cx + delta[:, 0] * width                                       # :45
    cy = cy + delta[:, 1] * height                                      # :46
    with np.errstate(over='ignore', invalid='ignore'):
        width = width * exp32(delta[:, 2])                              # :47
        height = height * exp32(delta[:, 3])                            # :48
        x1 = cx - F32(0.5) * width                                      # :50
        y1 = cy - F32(0.5) * height                                     # :51
        x2 = x1 + width                                                 # :52
        y2 = y1 + height                                                # :53
    return np.stack([x1, y1, x2, y2], axis=1).astype(np.float32)


# --------------------------------------------------------------------------------------
# utils/bbox_tf.py
# --------------------------------------------------------------------------------------
def area(boxes):
    """utils/bbox_tf.py:7-15 (+1 convention)."""
    b = f32(boxes)
    return (b[:, 3] - b[:, 1] + F32(1.0)) * (b[:, 2] - b[:, 0] + F32(1.0))


def pairwise_intersection(b1, b2):
    """utils/bbox_tf.py:18-34."""
    b1 = f32(b1)
    b2 = f32(b2)
    x_min1, y_min1, x_max1, y_max1 = [b1[:, i:i + 1] for i in range(4)]
    x_min2, y_min2, x_max2, y_max2 = [b2[:, i:i + 1] for i in range(4)]
    min_ymax = np.minimum(y_max1, y_max2.T)
    max_ymin = np.maximum(y_min1, y_min2.T)
    ih = np.maximum(F32(0.0), min_ymax - max_ymin + F32(1.0))
    min_xmax = np.minimum(x_max1, x_max2.T)
    max_xmin = np.maximum(x_min1, x_min2.T)
    iw = np.maximum(F32(0.0), min_xmax - max_xmin + F32(1.0))
    return ih * iw


def pairwise_iou(b1, b2):
    """utils/bbox_tf.py:37-56: 0 where intersection == 0 else inter / union."""
    inter = pairwise_intersection(b1, b2)
    a1 = area(b1)
    a2 = area(b2)
    unions = a1[:, None] + a2[None, :] - inter
    with np.errstate(divide='ignore', invalid='ignore'):
        q = inter / unions
    return np.where(inter == F32(0.0), F32(0.0), q).astype(np.float32)


def bboxes_clip_filter(boxes, min_value, max_height, max_width, min_edge=None):
    """utils/bbox_tf.py:59-84.  Returns (boxes, idx int64 ascending)."""
    b = f32(boxes).reshape(-1, 4)
    mv = F32(min_value)
    wmax = F32(max_width - 1)
    hmax = F32(max_height - 1)
    c0 = np.maximum(np.minimum(b[:, 0], wmax), mv)
    c1 = np.maximum(np.minimum(b[:, 1], hmax), mv)
    c2 = np.maximum(np.minimum(b[:, 2], wmax), mv)
    c3 = np.maximum(np.minimum(b[:, 3], hmax), mv)
    out = np.stack([c0, c1, c2, c3], axis=1)
    if min_edge is None:
        return out, np.arange(out.shape[0], dtype=np.int64)
    me = F32(min_edge)
    y_len = c2 - c0 + F32(1.0)          # sic: names swapped in the reference (:81-82)
    x_len = c3 - c1 + F32(1.0)
    idx = np.nonzero(np.logical_and(x_len >= me, y_len >= me))[0].astype(np.int64)
    return out[idx], idx


def bboxes_range_filter(anchors, max_height, max_width):
    """utils/bbox_tf.py:87-101."""
    a = f32(anchors)
    ok = ((a[:, 0] >= 0) & (a[:, 1] >= 0) &
          (a[:, 2] <= F32(max_width - 1)) & (a[:, 3] <= F32(max_height - 1)))
    return np.nonzero(ok)[0].astype(np.int64)


# --------------------------------------------------------------------------------------
# TensorFlow 1.x kernel restatements (third-party semantics; see module docstring)
# --------------------------------------------------------------------------------------
def _tf_iou_gt(bi, bj, thr):
    """TF r1.13 non_max_suppression_op.cc IOUGreaterThanThreshold: corner-sorted boxes,
    no +1, area<=0 never suppresses, strict '>' on a float32 quotient."""
    ymin_i = min(bi[0], bi[2]); xmin_i = min(bi[1], bi[3])
    ymax_i = max(bi[0], bi[2]); xmax_i = max(bi[1], bi[3])
    ymin_j = min(bj[0], bj[2]); xmin_j = min(bj[1], bj[3])
    ymax_j = max(bj[0], bj[2]); xmax_j = max(bj[1], bj[3])
    area_i = F32(F32(ymax_i - ymin_i) * F32(xmax_i - xmin_i))
    area_j = F32(F32(ymax_j - ymin_j) * F32(xmax_j - xmin_j))
    if area_i <= 0 or area_j <= 0:
        return False
    iymin = max(ymin_i, ymin_j); ixmin = max(xmin_i, xmin_j)
    iymax = min(ymax_i, ymax_j); ixmax = min(xmax_i, xmax_j)
    inter = F32(max(F32(iymax - iymin), F32(0.0)) * max(F32(ixmax - ixmin), F32(0.0)))
    iou = F32(inter / F32(F32(area_i + area_j) - inter))
    return bool(iou > thr)


def tf_non_max_suppression(boxes, scores, max_output_size, iou_threshold):
    """tf.image.non_max_suppression == NonMaxSuppressionV3 with score_threshold=-inf
    (call sites: model/region_proposal.py:74-76, model/prediction.py:146).
    Max-heap pop order (score desc, index asc); each candidate is tested against the kept
    boxes newest-first; stops at max_output_size.  Returns int32 indices in keep order."""
    boxes = f32(boxes).reshape(-1, 4)
    scores = f32(scores).reshape(-1)
    n = boxes.shape[0]
    thr = F32(iou_threshold)
    out_size = min(int(max_output_size), n)
    lowest = np.finfo(np.float32).min
    heap = [(-float(scores[i]), i) for i in range(n) if scores[i] > lowest]
    heapq.heapify(heap)
    selected = []
    while len(selected) < out_size and heap:
        _, i = heapq.heappop(heap)
        keep = True
        bi = boxes[i]
        for j in reversed(selected):
            if _tf_iou_gt(bi, boxes[j], thr):
                keep = False
                break
        if keep:
            selected.append(i)
    return np.asarray(selected, dtype=np.int32)


def tf_crop_and_resize(image, boxes, box_ind, crop_size):
    """tf.image.crop_and_resize, bilinear, extrapolation_value 0 (TF r1.13
    crop_and_resize_op.cc CPU functor; call sites model/roi_pooling.py:37,79,86,134).
    image [B,H,W,C] f32, boxes [R,4] normalised (y1,x1,y2,x2)."""
    image = f32(image)
    boxes = f32(boxes).reshape(-1, 4)
    _, H, W, C = image.shape
    ch, cw = int(crop_size[0]), int(crop_size[1])
    R = boxes.shape[0]
    out = np.zeros((R, ch, cw, C), dtype=np.float32)
    Hm1 = F32(H - 1)
    Wm1 = F32(W - 1)
    for b in range(R):
        y1, x1, y2, x2 = boxes[b]
        img = image[int(box_ind[b])]
        hs = F32(F32(F32(y2 - y1) * Hm1) / F32(ch - 1)) if ch > 1 else F32(0)
        ws = F32(F32(F32(x2 - x1) * Wm1) / F32(cw - 1)) if cw > 1 else F32(0)
        for y in range(ch):
            if ch > 1:
                in_y = F32(F32(y1 * Hm1) + F32(F32(y) * hs))
            else:
                in_y = F32(F32(F32(0.5) * F32(y1 + y2)) * Hm1)
            if in_y < 0 or in_y > Hm1:
                continue
            top = int(np.floor(in_y)); bot = int(np.ceil(in_y))
            y_lerp = F32(in_y - F32(top))
            for x in range(cw):
                if cw > 1:
                    in_x = F32(F32(x1 * Wm1) + F32(F32(x) * ws))
                else:
                    in_x = F32(F32(F32(0.5) * F32(x1 + x2)) * Wm1)
                if in_x < 0 or in_x > Wm1:
                    continue
                left = int(np.floor(in_x)); right = int(np.ceil(in_x))
                x_lerp = F32(in_x - F32(left))
                tl = img[top, left]; tr = img[top, right]
                bl = img[bot, left]; br = img[bot, right]
                t = tl + (tr - tl) * x_lerp
                bt = bl + (br - bl) * x_lerp
                out[b, y, x] = t + (bt - t) * y_lerp
    return out


def tf_max_pool_2x2(x):
    """Keras MaxPooling2D(padding='same') default 2x2/2 on an even map (roi_pooling.py:13,51)."""
    R, H, W, C = x.shape
    assert H % 2 == 0 and W % 2 == 0
    return x.reshape(R, H // 2, 2, W // 2, 2, C).max(axis=(2, 4))


def tf_avg_pool_2x2(x):
    """tf.nn.avg_pool 2x2/2 'SAME' on an even map (roi_pooling.py:154): Eigen sums the
    window in row-major order then divides by the count."""
    x = f32(x)
    s = ((x[:, 0::2, 0::2] + x[:, 0::2, 1::2]) + x[:, 1::2, 0::2]) + x[:, 1::2, 1::2]
    return (s / F32(4.0)).astype(np.float32)


def tf_top_k(values, k, sorted=True):
    """tf.nn.top_k CPU (TopKV2): (value desc, index asc).  With sorted=False the ORDER of
    the k results is unspecified in TF (heap order); the oracle returns the sorted order,
    callers compare as sets / after canonical ordering."""
    v = f32(values).reshape(-1)
    order = np.lexsort((np.arange(v.size), -v.astype(np.float64)))
    idx = order[:int(k)].astype(np.int32)
    return v[idx], idx


def tf_softmax(logits):
    """tf.nn.softmax CPU (softmax_op_functor.h): exp(x - max) * (1 / sum)."""
    x = f32(logits)
    m = x.max(axis=-1, keepdims=True)
    e = exp32(x - m)
    # Eigen reduces the class axis left to right for tiny inner dims
    s = e[..., 0].copy()
    for c in range(1, e.shape[-1]):
        s = s + e[..., c]
    inv = F32(1.0) / s
    return (e * inv[..., None]).astype(np.float32)


# --------------------------------------------------------------------------------------
# model/region_proposal.py
# --------------------------------------------------------------------------------------
def region_proposal(deltas, anchors, scores, image_shape, num_post_nms, nms_iou_threshold=0.7,
                    target_means=(0, 0, 0, 0), target_stds=(1, 1, 1, 1), return_idx=False):
    """model/region_proposal.py:55-81: decode -> clip (no filter) -> NMS over ALL anchors
    (pre-NMS top-k is commented out, :65-69) -> gather."""
    boxes = decode_bbox_with_mean_and_std(anchors, deltas, target_means, target_stds)   # :59
    boxes, _ = bboxes_clip_filter(boxes, 0, image_shape[0], image_shape[1])             # :63
    idx = tf_non_max_suppression(boxes, scores, num_post_nms, nms_iou_threshold)       # :74
    rois = boxes[idx]                                                                    # :81
    return (rois, idx) if return_idx else rois


# --------------------------------------------------------------------------------------
# model/roi_pooling.py
# --------------------------------------------------------------------------------------
def roi_pooling_crop_and_resize(feat, rois, extractor_stride, pool_size=7, max_pooling_flag=True):
    """model/roi_pooling.py:53-90 (RoiPoolingCropAndResize.call)."""
    feat = f32(feat)
    rois = f32(rois).reshape(-1, 4) / F32(extractor_stride)         # :64
    h, w = feat.shape[1:3]
    bb = np.stack([rois[:, 1] / F32(h - 1), rois[:, 0] / F32(w - 1),
                   rois[:, 3] / F32(h - 1), rois[:, 2] / F32(w - 1)], axis=1)   # :69-74
    ind = np.zeros(rois.shape[0], dtype=np.int32)
    if max_pooling_flag:
        crops = tf_crop_and_resize(feat, bb, ind, [2 * pool_size, 2 * pool_size])     # :79
        return tf_max_pool_2x2(crops)                                                 # :84
    return tf_crop_and_resize(feat, bb, ind, [pool_size, pool_size])                  # :86


def roi_pooling_crop_and_resize2(feat, rois, image_shape, pool_size=7):
    """model/roi_pooling.py:15-42 (RoiPoolingCropAndResize2.call, FPN variant: normalise
    by IMAGE size)."""
    feat = f32(feat)
    rois = f32(rois).reshape(-1, 4)
    h = F32(image_shape[0])
    w = F32(image_shape[1])
    bb = np.stack([rois[:, 1] / h, rois[:, 0] / w, rois[:, 3] / h, rois[:, 2] / w], axis=1)  # :30-35
    ind = np.zeros(rois.shape[0], dtype=np.int32)
    crops = tf_crop_and_resize(feat, bb, ind, [2 * pool_size, 2 * pool_size])         # :37
    return tf_max_pool_2x2(crops)                                                     # :42


def crop_and_resize_tp(image, boxes, box_ind, crop_size, pad_border=True):
    """model/roi_pooling.py:93-137 (tensorpack-style crop_and_resize helper)."""
    image = f32(image)
    boxes = f32(boxes).reshape(-1, 4)
    if pad_border:
        image = np.pad(image, [[0, 0], [1, 1], [1, 1], [0, 0]], mode='symmetric')     # :100
        boxes = boxes + F32(1)                                                        # :101
    H, W = image.shape[1:3]
    x0, y0, x1, y1 = boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 3]
    cs = F32(crop_size)
    spacing_w = (x1 - x0) / cs                                                        # :120
    spacing_h = (y1 - y0) / cs
    imh = F32(H - 1)
    imw = F32(W - 1)
    nx0 = (x0 + spacing_w / F32(2) - F32(0.5)) / imw                                  # :124
    ny0 = (y0 + spacing_h / F32(2) - F32(0.5)) / imh
    nw = spacing_w * F32(crop_size - 1) / imw                                         # :127
    nh = spacing_h * F32(crop_size - 1) / imh
    bb = np.stack([ny0, nx0, ny0 + nh, nx0 + nw], axis=1)                             # :130
    return tf_crop_and_resize(image, bb, box_ind, [crop_size, crop_size])


def roi_align(featuremap, boxes, resolution):
    """model/roi_pooling.py:140-155."""
    ind = np.zeros(np.asarray(boxes).reshape(-1, 4).shape[0], dtype=np.int32)
    ret = crop_and_resize_tp(featuremap, boxes, ind, resolution * 2)
    return tf_avg_pool_2x2(ret)


def roi_pooling_roi_align(feat, rois, extractor_stride, pool_size=7):
    """model/roi_pooling.py:164-177 (RoiPoolingRoiAlign.call)."""
    rois = f32(rois).reshape(-1, 4) / F32(extractor_stride)
    return roi_align(feat, rois, pool_size)


# --------------------------------------------------------------------------------------
# model/prediction.py
# --------------------------------------------------------------------------------------
def post_ops_prediction(roi_scores_softmax, roi_txtytwth, rois, image_shape,
                        target_means, target_stds, max_num_per_class=50, max_num_per_image=150,
                        nms_iou_threshold=0.3, score_threshold=0.05, extractor_stride=16,
                        num_classes=21):
    """model/prediction.py:103-163.  Output order: canonical (score desc, concat-position
    asc) -- TF's top_k(sorted=False) order is unspecified when k < n."""
    if target_stds is None:
        target_stds = [1, 1, 1, 1]
    if target_means is None:
        target_means = [0, 0, 0, 0]
    S = f32(roi_scores_softmax)
    D = f32(roi_txtytwth).reshape(S.shape[0], -1, 4)
    rois = f32(rois).reshape(-1, 4)
    res_scores, res_bboxes, res_cls = [], [], []
    for i in range(1, num_classes):                                                   # :135
        inds = np.nonzero(S[:, i] > F32(score_threshold))[0]                          # :136
        cls_score = S[inds, i]
        boxes = decode_bbox_with_mean_and_std(rois[inds], D[inds, i, :],
                                              target_means, target_stds)              # :138-140
        boxes, sel = bboxes_clip_filter(boxes, 0, image_shape[0], image_shape[1],
                                        extractor_stride)                             # :141-143
        cls_score = cls_score[sel]                                                    # :144
        keep = tf_non_max_suppression(boxes, cls_score, max_num_per_class, nms_iou_threshold)
        if keep.size == 0:                                                            # :147
            continue
        res_scores.append(cls_score[keep])
        res_bboxes.append(boxes[keep])
        res_cls.append(np.full(keep.shape, i, dtype=np.int32))
    if len(res_scores) == 0:
        return None, None, None                                                      # :153-154
    scores = np.concatenate(res_scores)
    bboxes = np.concatenate(res_bboxes)
    cls = np.concatenate(res_cls)
    _, final_idx = tf_top_k(scores, min(max_num_per_image, scores.size), sorted=False)  # :160
    return bboxes[final_idx], cls[final_idx], scores[final_idx]


def predict_after_roi(roi_scores_softmax, roi_txtytwth, rois, image_shape,
                      target_means, target_stds, max_num_per_class=5, max_num_per_image=5,
                      nms_iou_threshold=0.3, score_threshold=0.3, extractor_stride=16):
    """model/prediction.py:10-100 (unwired alternative; arg-max class per RoI)."""
    S = f32(roi_scores_softmax)
    D = f32(roi_txtytwth).reshape(S.shape[0], -1, 4)
    n = S.shape[0]
    class_ids = np.argmax(S, axis=1).astype(np.int32)                                 # :35
    class_scores = S[np.arange(n), class_ids]                                         # :39
    deltas = D[np.arange(n), class_ids]                                               # :41
    refined = decode_bbox_with_mean_and_std(rois, deltas, target_means, target_stds)  # :44
    refined, _ = bboxes_clip_filter(refined, 0, image_shape[0], image_shape[1], None)  # :46
    keep = np.nonzero((class_ids > 0) & (class_scores >= F32(score_threshold)))[0]    # :51-58
    pre_ids = class_ids[keep]
    pre_scores = class_scores[keep]
    pre_rois = refined[keep]
    uniq = []
    for c in pre_ids:                      # tf.unique: first-occurrence order (:65)
        if c not in uniq:
            uniq.append(int(c))
    nms_keep = []
    for c in uniq:                                                                   # :84-85
        ixs = np.nonzero(pre_ids == c)[0]
        ck = tf_non_max_suppression(pre_rois[ixs], pre_scores[ixs], max_num_per_class,
                                    nms_iou_threshold)
        nms_keep.append(keep[ixs[ck]])
    if len(nms_keep) == 0:
        return None, None, None
    nms_keep = np.concatenate(nms_keep)
    keep2 = np.intersect1d(keep, nms_keep)            # set_intersection -> ascending (:91-93)
    sc = class_scores[keep2]
    k = min(sc.size, max_num_per_image)
    _, top_ids = tf_top_k(sc, k, sorted=True)                                         # :97
    keep2 = keep2[top_ids]
    return refined[keep2], class_ids[keep2], class_scores[keep2]


# --------------------------------------------------------------------------------------
# caller glue (base_faster_rcnn_model.py / base_fpn_model.py)
# --------------------------------------------------------------------------------------
def rpn_fg_scores_frcnn(rpn_score, num_anchors):
    """model/faster_rcnn/base_faster_rcnn_model.py:149-152: channels = [A bg | A fg];
    fg[l,a] = softmax(s[l,a], s[l,A+a])[1]."""
    s = f32(rpn_score).reshape(-1, 2, num_anchors).transpose(0, 2, 1).reshape(-1, 2)
    p = tf_softmax(s).reshape(-1, num_anchors, 2).transpose(0, 2, 1).reshape(-1, 2 * num_anchors)
    return p[:, num_anchors:].reshape(-1).astype(np.float32)


def rpn_fg_scores_fpn(all_fpn_scores):
    """model/fpn/base_fpn_model.py:223 (+ :429 reshape [-1,2]): softmax(...)[:,1]."""
    return tf_softmax(f32(all_fpn_scores).reshape(-1, 2))[:, 1].astype(np.float32)


def assign_levels(all_rois, min_level=2, max_level=5):
    """model/fpn/base_fpn_model.py:303-324 (_assign_levels).  Returns (rois_list,
    concat index list int64, levels f32)."""
    r = f32(all_rois).reshape(-1, 4)
    h = np.maximum(F32(0.), r[:, 3] - r[:, 1])
    w = np.maximum(F32(0.), r[:, 2] - r[:, 0])
    log2 = log32(F32(2.))
    levels = np.floor(F32(4.) + log32(np.sqrt(w * h + F32(1e-8)) / F32(224.0)) / log2)   # :309
    levels = np.maximum(levels, F32(min_level))
    levels = np.minimum(levels, F32(max_level))
    rois_list, idx_list = [], []
    for i in range(min_level, max_level + 1):
        ii = np.nonzero(levels == F32(i))[0].astype(np.int64)
        rois_list.append(r[ii])
        idx_list.append(ii)
    return rois_list, np.concatenate(idx_list), levels


def fpn_roi_features(rois_list, p_list, image_shape, pool_size=7):
    """model/fpn/base_fpn_model.py:152-161 (_get_roi_features): per non-empty level,
    concatenated in level order."""
    feats = []
    for rois_k, p in zip(rois_list, p_list):
        if rois_k.shape[0] == 0:
            continue
        feats.append(roi_pooling_crop_and_resize2(p, rois_k, image_shape, pool_size))
    return np.concatenate(feats, axis=0)


# --------------------------------------------------------------------------------------
# model/fpn/resnet_fpn.py (neck, top-down merge) -- SURVEY 8(f) rank 3
# --------------------------------------------------------------------------------------
def tf_resize_bilinear_legacy(x, out_hw):
    """tf.image.resize_bilinear(x, size) of TF 1.x, align_corners=False (resize_bilinear_op.cc, SURVEY
    Appendix A): scale = in / out (float32); src = dst * scale; lo = floor(src); hi = min(lo + 1, in - 1);
    top = tl + (tr - tl) * xl; bottom = bl + (br - bl) * xl; out = top + (bottom - top) * yl.  x: [B,H,W,C]."""
    x = f32(x)
    B, H, W, Cc = x.shape
    oh, ow = int(out_hw[0]), int(out_hw[1])
    ys = F32(H) / F32(oh)
    xs = F32(W) / F32(ow)
    fy = (np.arange(oh, dtype=np.float32) * ys).astype(np.float32)
    fx = (np.arange(ow, dtype=np.float32) * xs).astype(np.float32)
    y0 = np.floor(fy).astype(np.int64)
    x0 = np.floor(fx).astype(np.int64)
    y1 = np.minimum(y0 + 1, H - 1)
    x1 = np.minimum(x0 + 1, W - 1)
    yl = (fy - y0.astype(np.float32)).astype(np.float32)[None, :, None, None]
    xl = (fx - x0.astype(np.float32)).astype(np.float32)[None, None, :, None]
    tl = x[:, y0][:, :, x0]
    tr = x[:, y0][:, :, x1]
    bl = x[:, y1][:, :, x0]
    br = x[:, y1][:, :, x1]
    top = (tl + ((tr - tl) * xl).astype(np.float32)).astype(np.float32)
    bot = (bl + ((br - bl) * xl).astype(np.float32)).astype(np.float32)
    return (top + ((bot - top) * yl).astype(np.float32)).astype(np.float32)


def fpn_topdown_merge(top, lateral):
    """model/fpn/resnet_fpn.py:385-398: Add([resize_bilinear(P_{k+1}, size(C_k)) * 0.5, lateral * 0.5])."""
    lateral = f32(lateral)
    up = tf_resize_bilinear_legacy(top, lateral.shape[1:3])
    return ((up * F32(0.5)).astype(np.float32) + (lateral * F32(0.5)).astype(np.float32)).astype(np.float32)


# --------------------------------------------------------------------------------------
# model/anchor_target.py, model/proposal_target.py -- the deterministic parts
# --------------------------------------------------------------------------------------
def anchor_target_labels(gt_bboxes, image_shape, all_anchors, pos_iou_threshold=0.7, neg_iou_threshold=0.3):
    """model/anchor_target.py:53-69 up to (not including) the random sub-sampling:
    -> (selected_anchor_idx, labels int32 in {-1,0,1} on the selected anchors, argmax_overlaps)."""
    idx = bboxes_range_filter(all_anchors, image_shape[0], image_shape[1])                 # :54
    anchors = f32(all_anchors)[idx]
    overlaps = pairwise_iou(anchors, gt_bboxes)                                            # :60
    argmax = np.argmax(overlaps, axis=1)                                                   # :61
    mx = overlaps.max(axis=1)                                                              # :62
    gt_max = overlaps.max(axis=0)                                                          # :63
    gt_argmax = np.nonzero(overlaps == gt_max[None, :])[0]                                 # :64
    labels = -np.ones(anchors.shape[0], np.int32)
    labels[mx < F32(neg_iou_threshold)] = 0                                                # :67
    labels[gt_argmax] = 1                                                                  # :68
    labels[mx >= F32(pos_iou_threshold)] = 1                                               # :69
    return idx, labels, argmax


def proposal_target_assign(rois, gt_bboxes, gt_labels, pos_iou_threshold=0.5, neg_iou_threshold=0.5):
    """model/proposal_target.py:55-63: -> (labels per RoI, gt_assignment, fg_inds, bg_inds)."""
    iou = pairwise_iou(rois, gt_bboxes)
    mx = iou.max(axis=1)
    ga = np.argmax(iou, axis=1)
    labels = np.asarray(gt_labels)[ga]
    fg = np.nonzero(mx >= F32(pos_iou_threshold))[0]
    bg = np.nonzero((mx < F32(pos_iou_threshold)) & (mx >= F32(neg_iou_threshold)))[0]
    return labels, ga, fg, bg


# --------------------------------------------------------------------------------------
# evaluation/detectron_pascal_evaluation_utils.py
# --------------------------------------------------------------------------------------
def voc_ap(rec, prec, use_07_metric=False):
    """evaluation/detectron_pascal_evaluation_utils.py:54-83.  07 metric: mean over the 11
    recall points t = 0,0.1,..,1.0 of max precision at recall >= t (0 if none).  Otherwise:
    area under the monotone precision envelope, summed where recall changes."""
    rec = np.asarray(rec, dtype=np.float64)
    prec = np.asarray(prec, dtype=np.float64)
    if use_07_metric:
        total = 0.0
        for t in np.arange(0., 1.1, 0.1):
            sel = rec >= t
            total = total + (np.max(prec[sel]) if sel.any() else 0) / 11.
        return total
    r = np.concatenate(([0.], rec, [1.]))
    p = np.concatenate(([0.], prec, [0.]))
    p = np.maximum.accumulate(p[::-1])[::-1]
    step = np.nonzero(r[1:] != r[:-1])[0]
    return np.sum((r[step + 1] - r[step]) * p[step + 1])


def voc_eval_arrays(dets, gt_boxes, gt_difficult, ovthresh=0.5, use_07_metric=True):
    """evaluation/detectron_pascal_evaluation_utils.py:86-222 for ONE class, on arrays instead of
    files: dets = list over images of [n,5] float arrays (x1,y1,x2,y2,score) in file order (image
    order, then the order of the image's rows), gt_boxes = list over images of [g,4], gt_difficult =
    list over images of bool [g].  Detections are visited by descending confidence
    (np.argsort(-confidence), :164 -- made stable here so that equal confidences keep file order),
    matched to the ground-truth box of highest overlap (+1 pixel convention, :181-196), a match above
    ovthresh is a TP the first time, FP afterwards, ignored on 'difficult' boxes (:198-206).
    -> (rec, prec, ap)."""
    npos = int(sum(int(np.sum(~np.asarray(d, dtype=bool))) for d in gt_difficult))
    image_ids, conf, bb = [], [], []
    for i, d in enumerate(dets):
        d = np.asarray(d, dtype=np.float64).reshape(-1, 5)
        for row in d:
            image_ids.append(i)
            conf.append(row[4])
            bb.append(row[:4])
    conf = np.asarray(conf, dtype=np.float64)
    bb = np.asarray(bb, dtype=np.float64).reshape(-1, 4)
    order = np.argsort(-conf, kind='stable')
    bb = bb[order]
    image_ids = [image_ids[k] for k in order]
    nd = len(image_ids)
    tp = np.zeros(nd)
    fp = np.zeros(nd)
    seen = [np.zeros(len(np.asarray(g).reshape(-1, 4)), dtype=bool) for g in gt_boxes]
    for d in range(nd):
        i = image_ids[d]
        g = np.asarray(gt_boxes[i], dtype=np.float64).reshape(-1, 4)
        diff = np.asarray(gt_difficult[i], dtype=bool).reshape(-1)
        ovmax, jmax = -np.inf, -1
        if g.size > 0:
            ixmin = np.maximum(g[:, 0], bb[d, 0])
            iymin = np.maximum(g[:, 1], bb[d, 1])
            ixmax = np.minimum(g[:, 2], bb[d, 2])
            iymax = np.minimum(g[:, 3], bb[d, 3])
            iw = np.maximum(ixmax - ixmin + 1., 0.)
            ih = np.maximum(iymax - iymin + 1., 0.)
            inters = iw * ih
            uni = ((bb[d, 2] - bb[d, 0] + 1.) * (bb[d, 3] - bb[d, 1] + 1.) +
                   (g[:, 2] - g[:, 0] + 1.) * (g[:, 3] - g[:, 1] + 1.) - inters)
            overlaps = inters / uni
            ovmax = np.max(overlaps)
            jmax = int(np.argmax(overlaps))
        if ovmax > ovthresh:
            if not diff[jmax]:
                if not seen[i][jmax]:
                    tp[d] = 1.
                    seen[i][jmax] = True
                else:
                    fp[d] = 1.
        else:
            fp[d] = 1.
    fp = np.cumsum(fp)
    tp = np.cumsum(tp)
    rec = tp / float(npos) if npos > 0 else tp * 0.0
    prec = tp / np.maximum(tp + fp, np.finfo(np.float64).eps)
    return rec, prec, voc_ap(rec, prec, use_07_metric)


# --------------------------------------------------------------------------------------
# evaluation/pascal_eval_files_utils.py  (the mAP-producing per-image loop)
# --------------------------------------------------------------------------------------
def eval_detect_image(scores, roi_txtytwth, rois, img_scale, raw_h, raw_w, num_classes=21,
                      score_threshold=0.0, iou_threshold=0.5, max_objects_per_class=50,
                      max_objects_per_image=50, target_means=None, target_stds=None, min_size=10):
    """evaluation/pascal_eval_files_utils.py:76-106 for one image, starting from the model's
    im_detect outputs (model/fpn/base_fpn_model.py:364-390): softmax scores [R,Ccls], raw deltas
    [R,4*Ccls] and rois in RESIZED-image pixels, divided by img_scale here as im_detect does (:390).
    Per class j = 1..num_classes-1: where(score > thr) -> decode -> clip to the RAW image with
    min_size filter -> NMS(max_objects_per_class) (:81-97); then the per-image cap: if more than
    max_objects_per_image detections survive, keep those with score >= the max_objects_per_image-th
    best score (:99-106; ties at the threshold are all kept).
    -> list (index = class id, entry 0 unused) of float32 [n,5] arrays (x1,y1,x2,y2,score)."""
    if target_stds is None:
        target_stds = [0.1, 0.1, 0.2, 0.2]
    if target_means is None:
        target_means = [0, 0, 0, 0]
    scores = f32(scores)
    R = scores.shape[0]
    D = f32(roi_txtytwth).reshape(R, -1, 4)
    rois = (f32(rois).reshape(-1, 4) / F32(img_scale)).astype(np.float32)
    raw_h = F32(raw_h)
    raw_w = F32(raw_w)
    out = [np.zeros((0, 5), np.float32) for _ in range(num_classes)]
    for j in range(1, num_classes):
        inds = np.nonzero(scores[:, j] > F32(score_threshold))[0]
        cls_scores = scores[inds, j]
        cls_boxes = decode_bbox_with_mean_and_std(rois[inds], D[inds, j, :], target_means, target_stds)
        cls_boxes, keep_idx = bboxes_clip_filter(cls_boxes, 0, raw_h, raw_w, min_size)
        cls_scores = cls_scores[keep_idx]
        keep = tf_non_max_suppression(cls_boxes, cls_scores, max_objects_per_class, iou_threshold)
        dets = np.hstack((cls_boxes, cls_scores[:, None])).astype(np.float32, copy=False)
        out[j] = dets[keep, :]
    if max_objects_per_image > 0:
        image_scores = np.hstack([out[j][:, -1] for j in range(1, num_classes)])
        if len(image_scores) > max_objects_per_image:
            image_thresh = np.sort(image_scores)[-max_objects_per_image]
            for j in range(1, num_classes):
                out[j] = out[j][out[j][:, -1] >= image_thresh, :]
    return out

"""Anchor generation -- counterpart of the reference's object_detection/utils/anchor_generator.py.

Same call surface (``generate_anchor_base``, ``generate_by_anchor_base_np``,
``generate_by_anchor_base_tf``, ``make_anchors``); the per-forward grid expansion runs as HIP
kernels (odet_anchors_shift / odet_anchors_fpn), the tiny A-row base tables stay on the host as
in the reference.  ``make_fpn_anchors`` is the native addition: all pyramid levels in ONE launch.
"""
import math

import numpy as np

from .. import ops
from ._device import current_device, to_gpu_f32

__all__ = ['generate_anchor_base', 'generate_by_anchor_base_np', 'generate_by_anchor_base_tf', 'make_anchors',
           'make_fpn_anchors', 'fpn_level_tables']


def generate_anchor_base(base_size=16, ratios=(0.5, 1, 2), scales=2 ** np.arange(3, 6)):
    """reference utils/anchor_generator.py:63-134.  Host float64 [A,4] (x1,y1,x2,y2), rows
    ratio-major / scale-minor, windows centred on the (0,0,base-1,base-1) reference box."""
    rows = []
    ctr = 0.5 * (float(base_size) - 1.0)
    area = float(base_size) * float(base_size)
    for r in [float(v) for v in np.asarray(ratios).reshape(-1)]:
        # numpy rounding (half to even) as in _ratio_enum (:118-120)
        w = float(np.round(np.sqrt(area / r)))
        h = float(np.round(w * r))
        for s in [float(v) for v in np.asarray(scales).reshape(-1)]:
            ws, hs = w * s, h * s
            rows.append([ctr - 0.5 * (ws - 1.0), ctr - 0.5 * (hs - 1.0),
                         ctr + 0.5 * (ws - 1.0), ctr + 0.5 * (hs - 1.0)])
    return np.asarray(rows, dtype=np.float64).reshape(-1, 4)


def generate_by_anchor_base_np(anchor_base, feat_stride, height, width):
    """reference utils/anchor_generator.py:23-43, reproduced literally (host numpy, as there):
    ``height``/``width`` are PIXEL sizes and the shift is laid out (y,x,y,x)."""
    base = np.asarray(anchor_base)
    ys = list(range(0, int(height), int(feat_stride)))
    xs = list(range(0, int(width), int(feat_stride)))
    out = np.empty((len(ys) * len(xs), base.shape[0], 4), dtype=np.float64)
    k = 0
    for y in ys:
        for x in xs:
            out[k] = base + np.asarray([y, x, y, x], dtype=np.float64)
            k += 1
    return out.reshape(-1, 4).astype(np.float32)


def generate_by_anchor_base_tf(anchor_base, feat_stride, height, width):
    """reference utils/anchor_generator.py:46-60.  ``height``/``width`` are feature-map cells
    (python ints or 0-d tensors).  Returns a float32 GPU tensor [height*width*A, 4]."""
    base = to_gpu_f32(anchor_base)
    return ops.anchors_shift(base, int(feat_stride), int(height), int(width))


def _wh_table(base_anchor_size, anchor_scales, anchor_ratios):
    """enum_scales + enum_ratios (reference :165-178) in float32, including the (hs, ws) swap:
    w = S*s*sqrt(r), h = (S*s)/sqrt(r); ratio-major / scale-minor."""
    side = np.float32(base_anchor_size) * np.asarray(anchor_scales, dtype=np.float32).reshape(-1)
    sr = np.sqrt(np.asarray(anchor_ratios, dtype=np.float32).reshape(-1))
    w = (side[None, :] * sr[:, None]).reshape(-1)
    h = (side[None, :] / sr[:, None]).reshape(-1)
    return np.stack([w, h], axis=1).astype(np.float32)


def make_anchors(base_anchor_size, anchor_scales, anchor_ratios, featuremap_height, featuremap_width, stride,
                 name='make_anchors'):
    """reference utils/anchor_generator.py:137-162.  featuremap_height/width may be floats (the
    FPN caller passes tf.ceil(H/stride) as float); tf.range(float limit) yields ceil(limit)
    cells."""
    fh = int(math.ceil(float(featuremap_height)))
    fw = int(math.ceil(float(featuremap_width)))
    wh = _wh_table(base_anchor_size, anchor_scales, anchor_ratios)[None]
    return ops.anchors_fpn([fh], [fw], [int(stride)], wh, current_device())


def fpn_level_tables(image_shape, anchor_stride_list, base_anchor_size_list, anchor_scales, anchor_ratios):
    """Host tables of reference model/fpn/base_fpn_model.py:163-186 (_get_anchors): per level
    ceil(H/stride), ceil(W/stride) and the float32 (w, h) table of make_anchors."""
    fh = [int(math.ceil(image_shape[0] / s)) for s in anchor_stride_list]
    fw = [int(math.ceil(image_shape[1] / s)) for s in anchor_stride_list]
    wh = np.stack([_wh_table(b, anchor_scales, anchor_ratios) for b in base_anchor_size_list], axis=0)
    return fh, fw, wh


def make_fpn_anchors(image_shape, anchor_stride_list, base_anchor_size_list, anchor_scales, anchor_ratios):
    """All pyramid levels of reference model/fpn/base_fpn_model.py:163-186 (_get_anchors) in one
    launch; levels concatenated in list order."""
    fh = [int(math.ceil(image_shape[0] / s)) for s in anchor_stride_list]
    fw = [int(math.ceil(image_shape[1] / s)) for s in anchor_stride_list]
    wh = np.stack([_wh_table(b, anchor_scales, anchor_ratios) for b in base_anchor_size_list], axis=0)
    return ops.anchors_fpn(fh, fw, [int(s) for s in anchor_stride_list], wh, current_device())

"""IoU / clip / range filters -- counterpart of the reference's object_detection/utils/bbox_tf.py.

Index outputs are int64 ascending (tf.where); the reference's ``min_edge=None`` branch returns
``tf.range`` (int32) -- int64 is used for both branches here.
"""
import torch

from .. import ops

__all__ = ['pairwise_iou', 'bboxes_clip_filter', 'bboxes_range_filter']


def pairwise_iou(boxlist1, boxlist2):
    """reference utils/bbox_tf.py:37-56 (+1 pixel convention; 0 where intersection == 0)."""
    return ops.pairwise_iou(boxlist1, boxlist2)


def bboxes_clip_filter(rpn_proposals, min_value, max_height, max_width, min_edge=None):
    """reference utils/bbox_tf.py:59-84 -> (boxes, idx)."""
    if min_edge is None:
        out = ops.clip(rpn_proposals, min_value, max_height, max_width)
        return out, torch.arange(out.shape[0], dtype=torch.int64, device=out.device)
    boxes, idx, cnt = ops.clip_filter(rpn_proposals, min_value, max_height, max_width, min_edge)
    m = int(cnt.item())     # dynamic output shape -> one host sync, as tf.where implies
    return boxes[:m], idx[:m]


def bboxes_range_filter(anchors, max_height, max_width):
    """reference utils/bbox_tf.py:87-101 -> int64 indices of anchors fully inside the image."""
    idx, cnt = ops.range_filter(anchors, max_height, max_width)
    return idx[:int(cnt.item())]

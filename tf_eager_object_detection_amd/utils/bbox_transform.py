"""Box <-> delta transforms -- counterpart of the reference's
object_detection/utils/bbox_transform.py (HIP kernels odet_encode / odet_decode)."""
from .. import ops

__all__ = ['encode_bbox_with_mean_and_std', 'decode_bbox_with_mean_and_std']


def encode_bbox_with_mean_and_std(src_bbox, dst_bbox, target_means, target_stds):
    """reference utils/bbox_transform.py:4-29.  [M,4] x [M,4] GPU tensors -> [M,4] deltas."""
    return ops.encode(src_bbox, dst_bbox, target_means, target_stds)


def decode_bbox_with_mean_and_std(anchors, bboxes_txtytwth, target_means, target_stds):
    """reference utils/bbox_transform.py:32-55.  Note x2 = x1 + w (no -1), no clamp on dw/dh."""
    return ops.decode(anchors, bboxes_txtytwth, target_means, target_stds)

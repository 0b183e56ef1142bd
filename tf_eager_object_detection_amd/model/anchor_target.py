"""RPN training targets -- counterpart of the reference's model/anchor_target.py (AnchorTarget, :7-107).

SURVEY.md section 8(f) rank 4: the heavy parts run on the path's HIP kernels (bboxes_range_filter = a8,
pairwise_iou = a9, encode_bbox_with_mean_and_std = a6); the reductions and index bookkeeping around them are a
handful of PyTorch-ROCm tensor ops on the GPU.  The reference samples with tf.random_shuffle; the stream of a
torch generator stands in for it, so parity is exact up to the sampling step (`labels_before_sampling`) and
distribution-level after it (counts, subset relations)."""
import torch

from ..utils.bbox_tf import bboxes_range_filter, pairwise_iou
from ..utils.bbox_transform import encode_bbox_with_mean_and_std

__all__ = ['AnchorTarget']


def _unmap(data, count, inds, fill=0.0):
    """anchor_target.py:110-125: results on the filtered anchors scattered back over all anchors (float32)."""
    shape = (count,) + tuple(data.shape[1:])
    ret = torch.full(shape, float(fill), dtype=torch.float32, device=data.device)
    ret[inds] = data.to(torch.float32)
    return ret


class AnchorTarget:
    """Same constructor and call signature as the reference's keras model."""

    def __init__(self, pos_iou_threshold=0.7, neg_iou_threshold=0.3, total_num_samples=256, max_pos_samples=128,
                 target_means=None, target_stds=None, generator=None):
        self._pos_iou_threshold = pos_iou_threshold
        self._neg_iou_threshold = neg_iou_threshold
        self._total_num_samples = total_num_samples
        self._max_pos_samples = max_pos_samples
        self._target_means = [0, 0, 0, 0] if target_means is None else target_means
        self._target_stds = [1, 1, 1, 1] if target_stds is None else target_stds
        self._generator = generator          # torch.Generator on the GPU (None: the default stream)

    def labels_before_sampling(self, gt_bboxes, image_shape, all_anchors):
        """The deterministic part (:53-72): -> (selected_anchor_idx int64, anchors, labels int32 in {-1,0,1},
        argmax_overlaps int64) on the anchors inside the image."""
        idx = bboxes_range_filter(all_anchors, image_shape[0], image_shape[1])             # :54
        anchors = all_anchors[idx]
        overlaps = pairwise_iou(anchors, gt_bboxes)                                        # :60 [anchors, gt]
        max_overlaps, argmax_overlaps = overlaps.max(dim=1)                                # :61-62 (first maximum)
        gt_max = overlaps.max(dim=0).values                                                # :63
        gt_argmax = (overlaps == gt_max.unsqueeze(0)).nonzero()[:, 0]                      # :64
        labels = torch.full((anchors.shape[0],), -1, dtype=torch.int32, device=anchors.device)
        labels[max_overlaps < self._neg_iou_threshold] = 0                                 # :67
        labels[gt_argmax] = 1                                                              # :68
        labels[max_overlaps >= self._pos_iou_threshold] = 1                                # :69
        return idx, anchors, labels, argmax_overlaps

    def _shuffle(self, t):
        return t[torch.randperm(t.numel(), device=t.device, generator=self._generator)]

    def __call__(self, inputs, training=None, mask=None):
        gt_bboxes, image_shape, all_anchors = inputs
        total = all_anchors.shape[0]
        idx, anchors, labels, argmax_overlaps = self.labels_before_sampling(gt_bboxes, image_shape, all_anchors)
        fg = (labels == 1).nonzero()[:, 0]                                                 # :72
        if fg.numel() > self._max_pos_samples:                                             # :73-77
            labels[self._shuffle(fg)[self._max_pos_samples:]] = -1
        num_bg = self._total_num_samples - int((labels == 1).sum().item())                 # :78
        bg = (labels == 0).nonzero()[:, 0]
        if bg.numel() > num_bg:                                                            # :80-84
            labels[self._shuffle(bg)[num_bg:]] = -1
        targets = encode_bbox_with_mean_and_std(anchors, gt_bboxes[argmax_overlaps], target_means=self._target_means,
                                                target_stds=self._target_stds)             # :88-90
        inside = torch.zeros((anchors.shape[0], 4), dtype=torch.float32, device=anchors.device)
        inside[labels == 1] = 1.0                                                          # :93-95
        outside = torch.zeros_like(inside)
        num_examples = (labels >= 0).sum().to(torch.float32)                               # :99
        outside[labels >= 0] = 1.0 / num_examples                                          # :100-101
        return (_unmap(labels, total, idx, -1), _unmap(targets, total, idx, 0),
                _unmap(inside, total, idx, 0), _unmap(outside, total, idx, 0))             # :104-107

    call = __call__

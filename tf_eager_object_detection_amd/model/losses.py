"""Losses -- counterpart of the reference's model/losses.py (:4-28), on torch tensors (autograd-capable)."""
import torch
import torch.nn.functional as F

__all__ = ['cls_loss', 'smooth_l1_loss']


def cls_loss(logits, labels, weight=1):
    """tf.losses.sparse_softmax_cross_entropy(logits, labels, weights=weight) with its default reduction
    (SUM_BY_NONZERO_WEIGHTS): weighted sum / number of non-zero weights."""
    ce = F.cross_entropy(logits, labels.to(torch.int64), reduction='none')
    w = torch.as_tensor(weight, dtype=ce.dtype, device=ce.device)
    w = w.expand_as(ce) if w.dim() > 0 else w.expand(ce.shape)
    nz = (w != 0).sum().clamp_min(1).to(ce.dtype)
    return (ce * w).sum() / nz


def smooth_l1_loss(bbox_pred, bbox_targets, bbox_inside_weights, bbox_outside_weights, sigma=1.0, dim=(1,)):
    """losses.py:16-28: smooth L1 with the py-faster-rcnn sigma, summed over `dim`, mean over the rest."""
    sigma_2 = sigma ** 2
    in_box_diff = bbox_inside_weights * (bbox_pred - bbox_targets)
    abs_in = in_box_diff.abs()
    sign = (abs_in < 1.0 / sigma_2).to(in_box_diff.dtype).detach()
    in_loss = in_box_diff.pow(2) * (sigma_2 / 2.0) * sign + (abs_in - 0.5 / sigma_2) * (1.0 - sign)
    return (bbox_outside_weights * in_loss).sum(dim=tuple(dim)).mean()

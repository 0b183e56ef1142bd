"""RoI-head training targets -- counterpart of the reference's model/proposal_target.py (ProposalTarget, :8-124).

SURVEY.md section 8(f) rank 4: pairwise_iou (a9) and encode_bbox_with_mean_and_std (a6) are the path's HIP
kernels; sampling uses a torch generator where the reference uses tf.random_shuffle / np.random.choice, so parity
is exact on the assignment (`assign`) and distribution-level on the sampled batch."""
import torch

from ..utils.bbox_tf import pairwise_iou
from ..utils.bbox_transform import encode_bbox_with_mean_and_std

__all__ = ['ProposalTarget']


class ProposalTarget:
    """Same constructor and call signature as the reference's keras model.

    reference_row_labels=True keeps a quirk of the reference: the class column a positive row's targets and
    inside weights are written to is taken from `labels[row]` (:96, :113 -- the label of RoI number `row` of the
    INPUT, not of the sampled foreground RoI in that row); False uses the sampled RoI's own label."""

    def __init__(self, num_classes=21, pos_iou_threshold=0.5, neg_iou_threshold=0.5, total_num_samples=128,
                 max_pos_samples=32, target_means=None, target_stds=None, generator=None, reference_row_labels=True):
        self._num_classes = num_classes
        self._pos_iou_threshold = pos_iou_threshold
        self._neg_iou_threshold = neg_iou_threshold
        self._total_num_samples = total_num_samples
        self._max_pos_samples = max_pos_samples
        self._target_means = [0, 0, 0, 0] if target_means is None else target_means
        self._target_stds = [1, 1, 1, 1] if target_stds is None else target_stds
        self._generator = generator
        self._reference_row_labels = reference_row_labels

    def assign(self, rois, gt_bboxes, gt_labels):
        """The deterministic part (:55-63): -> (labels of the best-overlapping gt per RoI, gt_assignment int64,
        fg_inds, bg_inds)."""
        iou = pairwise_iou(rois, gt_bboxes)                                                # :55
        max_overlaps, gt_assignment = iou.max(dim=1)                                       # :56-57
        labels = gt_labels[gt_assignment]                                                  # :58
        fg = (max_overlaps >= self._pos_iou_threshold).nonzero()[:, 0]                     # :61
        bg = ((max_overlaps < self._pos_iou_threshold) &
              (max_overlaps >= self._neg_iou_threshold)).nonzero()[:, 0]                   # :62-63
        return labels, gt_assignment, fg, bg

    def _shuffle(self, t):
        return t[torch.randperm(t.numel(), device=t.device, generator=self._generator)]

    def __call__(self, inputs, training=None, mask=None):
        rois, gt_bboxes, gt_labels = inputs
        labels, gt_assignment, fg, bg = self.assign(rois, gt_bboxes, gt_labels)
        if fg.numel() > self._max_pos_samples:                                             # :66-67
            fg = self._shuffle(fg)[:self._max_pos_samples]
        want_bg = self._total_num_samples - fg.numel()
        if bg.numel() > want_bg:                                                           # :68-70
            bg = self._shuffle(bg)[:want_bg]
        elif bg.numel() < want_bg:                                                         # :73-76 np.random.choice
            if bg.numel() == 0:
                raise ValueError('no background RoI to sample from (the reference fails here as well)')
            pick = torch.randint(0, bg.numel(), (want_bg,), device=bg.device, generator=self._generator)
            bg = bg[pick]
        keep = torch.cat([fg, bg])                                                         # :80
        final_rois = rois[keep]
        final_labels = labels[keep].clone()
        final_labels[fg.numel():] = 0                                                      # :84-85
        nfg, nk = fg.numel(), keep.numel()
        inside = torch.zeros((nk, self._num_classes, 4), dtype=torch.float32, device=rois.device)
        targets = torch.zeros_like(inside)
        if nfg > 0:
            rows = torch.arange(nfg, device=rois.device)
            cols = (labels[:nfg] if self._reference_row_labels else labels[fg]).to(torch.int64)   # :96 / :113
            inside[rows, cols] = 1.0
            enc = encode_bbox_with_mean_and_std(final_rois[:nfg], gt_bboxes[gt_assignment[fg]],
                                                target_means=self._target_means, target_stds=self._target_stds)
            targets[rows, cols] = enc                                                      # :103-113
        inside = inside.reshape(nk, self._num_classes * 4)
        targets = targets.reshape(nk, self._num_classes * 4)
        return final_rois, final_labels, targets, inside, torch.ones_like(inside)          # :118-124

    call = __call__

"""RPN proposal layer -- counterpart of the reference's object_detection/model/region_proposal.py.

decode -> clip -> exact greedy NMS over ALL anchors -> gather, as one C-ABI call
(odet_region_proposal: fused decode+clip kernel, radix sort, bit-matrix NMS)."""
import torch

from .. import ops

__all__ = ['RegionProposal']


class RegionProposal(torch.nn.Module):
    """reference model/region_proposal.py:11-81.  Same constructor arguments; ``num_anchors``
    and ``num_pre_nms_*`` are stored and unused exactly as there (the pre-NMS top-k is commented
    out at :65-69)."""

    def __init__(self, num_anchors=9, num_pre_nms_train=12000, num_post_nms_train=2000, num_pre_nms_test=6000,
                 num_post_nms_test=300, nms_iou_threshold=0.7, target_means=None, target_stds=None):
        super().__init__()
        self._num_anchors = num_anchors
        self._num_pre_nms_train = num_pre_nms_train
        self._num_post_nms_train = num_post_nms_train
        self._num_pre_nms_test = num_pre_nms_test
        self._num_post_nms_test = num_post_nms_test
        self._nms_iou_threshold = nms_iou_threshold
        self._target_means = [0, 0, 0, 0] if target_means is None else target_means
        self._target_stds = [1, 1, 1, 1] if target_stds is None else target_stds

    def padded(self, inputs, training=None):
        """Sync-free form: (rois [K,4] padded, kept anchor idx int32 [K], count int32[1] on device)."""
        bboxes_txtytwth, anchors, scores, image_shape = inputs
        num_post_nms = self._num_post_nms_train if training else self._num_post_nms_test   # :73
        return ops.region_proposal(bboxes_txtytwth, anchors, scores, image_shape, num_post_nms,
                                   self._nms_iou_threshold, self._target_means, self._target_stds)

    @torch.no_grad()
    def forward(self, inputs, training=None, mask=None):
        """inputs = (deltas [N,4], anchors [N,4], fg scores [N], image_shape [H, W]) ->
        rois [K' <= num_post_nms, 4] in NMS pick order (descending score), no gradient (:81)."""
        rois, _, cnt = self.padded(inputs, training)
        return rois[:int(cnt.item())]

    call = forward

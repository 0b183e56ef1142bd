"""Detection post-processing -- counterpart of the reference's object_detection/model/prediction.py."""
import torch

from .. import ops
from ..utils.bbox_tf import bboxes_clip_filter
from ..utils.bbox_transform import decode_bbox_with_mean_and_std

__all__ = ['post_ops_prediction']


def post_ops_prediction_padded(roi_scores_softmax, roi_txtytwth, rois, image_shape, target_means, target_stds,
                               max_num_per_class=50, max_num_per_image=150, nms_iou_threshold=0.3,
                               score_threshold=0.05, extractor_stride=16, num_classes=21, count_dev=None):
    """Sync-free form: (boxes [max_per_image,4], labels int32, scores, count int32[1] on device)."""
    if target_stds is None:
        target_stds = [1, 1, 1, 1]
    if target_means is None:
        target_means = [0, 0, 0, 0]
    return ops.post_ops(roi_scores_softmax, roi_txtytwth, rois, image_shape, target_means, target_stds,
                        max_num_per_class, max_num_per_image, nms_iou_threshold, score_threshold,
                        extractor_stride, num_classes, count_dev=count_dev)


@torch.no_grad()
def post_ops_prediction(roi_scores_softmax, roi_txtytwth, rois, image_shape, target_means, target_stds,
                        max_num_per_class=50, max_num_per_image=150, nms_iou_threshold=0.3, score_threshold=0.05,
                        extractor_stride=16, num_classes=21):
    """reference model/prediction.py:103-163.  Returns (boxes [M,4], labels int32 [M], scores [M])
    or (None, None, None) when nothing survives (:153-154).  All foreground classes run in one
    launch; the M results come back sorted by (score desc), which is one valid order of the
    reference's ``top_k(sorted=False)``."""
    boxes, labels, scores, cnt = post_ops_prediction_padded(
        roi_scores_softmax, roi_txtytwth, rois, image_shape, target_means, target_stds, max_num_per_class,
        max_num_per_image, nms_iou_threshold, score_threshold, extractor_stride, num_classes)
    m = int(cnt.item())
    if m == 0:
        return None, None, None
    return boxes[:m], labels[:m], scores[:m]


@torch.no_grad()
def predict_after_roi(roi_scores_softmax, roi_txtytwth, rois, image_shape, target_means, target_stds,
                      max_num_per_class=5, max_num_per_image=5, nms_iou_threshold=0.3, score_threshold=0.3,
                      extractor_stride=16):
    """reference model/prediction.py:10-100 (unwired alternative: arg-max class per RoI, per
    present class NMS, top-k sorted).  Index plumbing uses torch ops; decode / clip / NMS are the
    HIP kernels."""
    S = roi_scores_softmax.float()
    n = S.shape[0]
    D = roi_txtytwth.float().reshape(n, -1, 4)
    class_ids = torch.argmax(S, dim=1).to(torch.int32)                                    # :35
    ar = torch.arange(n, device=S.device)
    class_scores = S[ar, class_ids.long()]                                                # :39
    deltas = D[ar, class_ids.long()].contiguous()                                         # :41
    refined = decode_bbox_with_mean_and_std(rois, deltas, target_means, target_stds)      # :44
    refined, _ = bboxes_clip_filter(refined, 0, image_shape[0], image_shape[1], None)     # :46
    keep = torch.nonzero((class_ids > 0) & (class_scores >= score_threshold)).reshape(-1)  # :51-58
    pre_ids = class_ids[keep]
    pre_scores = class_scores[keep]
    pre_rois = refined[keep]
    uniq = []
    for c in pre_ids.tolist():            # tf.unique keeps first-occurrence order (:65)
        if c not in uniq:
            uniq.append(c)
    nms_keep = []
    for c in uniq:                                                                        # :84-85
        ixs = torch.nonzero(pre_ids == c).reshape(-1)
        idx, cnt = ops.nms(pre_rois[ixs].contiguous(), pre_scores[ixs].contiguous(), max_num_per_class,
                           nms_iou_threshold)
        ck = idx[:int(cnt.item())].long()
        nms_keep.append(keep[ixs[ck]])
    if len(nms_keep) == 0:
        return None, None, None                                                           # :87-88
    nms_keep = torch.cat(nms_keep)
    keep2 = torch.sort(nms_keep).values          # set_intersection(keep, nms_keep) -> ascending (:91-93)
    sc = class_scores[keep2]
    k = min(sc.shape[0], max_num_per_image)
    # tf.nn.top_k(sorted=True): value desc, index asc (:97)
    order = torch.sort(sc, descending=True, stable=True).indices[:k]
    keep2 = keep2[order]
    return refined[keep2], class_ids[keep2], class_scores[keep2]

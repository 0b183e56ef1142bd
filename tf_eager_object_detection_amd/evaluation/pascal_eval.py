"""PASCAL-VOC style evaluation of the detection path -- counterpart of the reference's
object_detection/evaluation/pascal_eval_files_utils.py (per-image detection loop, result files) and
object_detection/evaluation/detectron_pascal_evaluation_utils.py (voc_ap / voc_eval).

The per-image loop (score threshold -> decode on rois / img_scale -> clip to the raw image with the
min_size filter -> per-class NMS -> per-image score cap) runs on the GPU as ONE C-ABI call
(odet_eval_detect: the post-ops kernels with the eval arguments, all classes in one launch, no host
sync per class).  AP bookkeeping is host numpy, as it is in the reference.
"""
import numpy as np
import torch

from .. import _lib as L
from .. import ops

__all__ = ['im_detect_outputs', 'detect_image', 'voc_ap', 'voc_eval_arrays', 'evaluate_detections',
           'write_voc_results_file']


def im_detect_outputs(roi_score_logits, roi_txtytwth, rois):
    """What BaseFPN.im_detect / BaseFasterRcnn.im_detect return before the division by img_scale
    (model/fpn/base_fpn_model.py:380-390, model/faster_rcnn/base_faster_rcnn_model.py:300-306):
    softmax over the class logits, the raw deltas [R, 4*Ccls], the RoIs.  detect_image() applies
    `rois / img_scale` inside the kernel."""
    return torch.softmax(roi_score_logits.float(), dim=-1), roi_txtytwth, rois


def detect_image(scores, roi_txtytwth, rois, img_scale, raw_h, raw_w, num_classes=21, score_threshold=0.0,
                 iou_threshold=0.5, max_objects_per_class=50, max_objects_per_image=50, target_means=None,
                 target_stds=None, min_size=10, count_dev=None):
    """reference evaluation/pascal_eval_files_utils.py:76-106 for one image.
    scores [R,Ccls] softmax, roi_txtytwth [R,4*Ccls] or [R,Ccls,4], rois [R,4] in resized-image
    pixels (GPU tensors).  -> list (index = class id, entry 0 unused) of float32 numpy [n,5] arrays
    (x1,y1,x2,y2,score), exactly the reference's all_boxes[j][i]."""
    if target_stds is None:
        target_stds = [0.1, 0.1, 0.2, 0.2]
    if target_means is None:
        target_means = [0, 0, 0, 0]
    scores = L.f32c(scores, 'scores')
    if scores.dim() != 2:
        raise ValueError('scores must be [num_rois, num_classes]')
    R, Ccls = scores.shape
    deltas = L.f32c(roi_txtytwth, 'roi_txtytwth')
    if deltas.numel() != R * Ccls * 4:
        raise ValueError('roi_txtytwth must hold [num_rois, num_classes, 4] values')
    rois = ops._boxes(rois, 'rois')
    if rois.shape[0] != R:
        raise ValueError('rois has %d rows for %d score rows' % (rois.shape[0], R))
    if num_classes > Ccls:
        raise ValueError('num_classes %d exceeds the %d score columns' % (num_classes, Ccls))
    cap = max((num_classes - 1) * int(max_objects_per_class), 1)
    dev = scores.device
    ob = torch.empty((cap, 4), dtype=torch.float32, device=dev)
    ol = torch.empty(cap, dtype=torch.int32, device=dev)
    os_ = torch.empty(cap, dtype=torch.float32, device=dev)
    cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    nb = L.lib().odet_post_ops_workspace_bytes(int(num_classes), int(max_objects_per_class))
    ws = L.workspace(nb, dev)
    L.call('odet_eval_detect', L.dptr(scores), L.dptr(deltas), L.dptr(rois), R, L.dptr(count_dev), Ccls,
           int(num_classes), float(img_scale), float(raw_h), float(raw_w), L.host4(target_means, 'target_means'),
           L.host4(target_stds, 'target_stds'), int(max_objects_per_class), int(max_objects_per_image),
           float(iou_threshold), float(score_threshold), float(min_size), L.dptr(ob), L.dptr(ol), L.dptr(os_),
           L.dptr(cnt), L.dptr(ws), nb, L.stream())
    m = int(cnt.item())                       # the reference converts every class to numpy here (:92-96)
    boxes = ob[:m].cpu().numpy()
    labels = ol[:m].cpu().numpy()
    sc = os_[:m].cpu().numpy()
    out = [np.zeros((0, 5), np.float32) for _ in range(num_classes)]
    for j in range(1, num_classes):
        sel = labels == j
        if sel.any():
            out[j] = np.hstack((boxes[sel], sc[sel][:, None])).astype(np.float32, copy=False)
    return out


def write_voc_results_file(path, image_ids, dets_per_image):
    """reference evaluation/pascal_eval_files_utils.py:109-122 for one class: one line per detection,
    `image_id score x1+1 y1+1 x2+1 y2+1` (VOCdevkit 1-based pixels), 3 / 1 decimals."""
    with open(path, 'wt') as f:
        for index, dets in zip(image_ids, dets_per_image):
            dets = np.asarray(dets)
            for k in range(dets.shape[0]):
                f.write('{:s} {:.3f} {:.1f} {:.1f} {:.1f} {:.1f}\n'.format(
                    str(index), dets[k, -1], dets[k, 0] + 1, dets[k, 1] + 1, dets[k, 2] + 1, dets[k, 3] + 1))


def voc_ap(rec, prec, use_07_metric=False):
    """reference evaluation/detectron_pascal_evaluation_utils.py:54-83."""
    rec = np.asarray(rec, dtype=np.float64)
    prec = np.asarray(prec, dtype=np.float64)
    if use_07_metric:
        ap = 0.
        for t in np.arange(0., 1.1, 0.1):
            p = np.max(prec[rec >= t]) if np.sum(rec >= t) > 0 else 0
            ap = ap + p / 11.
        return ap
    mrec = np.concatenate(([0.], rec, [1.]))
    mpre = np.concatenate(([0.], prec, [0.]))
    for i in range(mpre.size - 1, 0, -1):
        mpre[i - 1] = np.maximum(mpre[i - 1], mpre[i])
    i = np.where(mrec[1:] != mrec[:-1])[0]
    return np.sum((mrec[i + 1] - mrec[i]) * mpre[i + 1])


def voc_eval_arrays(dets, gt_boxes, gt_difficult, ovthresh=0.5, use_07_metric=True):
    """reference evaluation/detectron_pascal_evaluation_utils.py:86-222 for one class on arrays:
    dets[i] = [n,5] (x1,y1,x2,y2,score) of image i, gt_boxes[i] = [g,4], gt_difficult[i] = bool [g].
    Detections in descending confidence (equal confidences keep image order), +1-pixel IoU, first
    match above ovthresh is a TP, later ones FP, 'difficult' boxes ignored.  -> (rec, prec, ap)."""
    npos = 0
    for d in gt_difficult:
        npos += int(np.sum(~np.asarray(d, dtype=bool)))
    ids = np.concatenate([np.full(len(np.asarray(d).reshape(-1, 5)), i, dtype=np.int64) for i, d in enumerate(dets)]
                         or [np.zeros(0, np.int64)])
    allb = np.concatenate([np.asarray(d, dtype=np.float64).reshape(-1, 5) for d in dets] or [np.zeros((0, 5))])
    order = np.argsort(-allb[:, 4], kind='stable')
    allb, ids = allb[order], ids[order]
    nd = allb.shape[0]
    tp, fp = np.zeros(nd), np.zeros(nd)
    taken = [np.zeros(len(np.asarray(g).reshape(-1, 4)), dtype=bool) for g in gt_boxes]
    for d in range(nd):
        i = int(ids[d])
        g = np.asarray(gt_boxes[i], dtype=np.float64).reshape(-1, 4)
        hard = np.asarray(gt_difficult[i], dtype=bool).reshape(-1)
        bb = allb[d, :4]
        ovmax, jmax = -np.inf, -1
        if g.size > 0:
            iw = np.maximum(np.minimum(g[:, 2], bb[2]) - np.maximum(g[:, 0], bb[0]) + 1., 0.)
            ih = np.maximum(np.minimum(g[:, 3], bb[3]) - np.maximum(g[:, 1], bb[1]) + 1., 0.)
            inters = iw * ih
            uni = ((bb[2] - bb[0] + 1.) * (bb[3] - bb[1] + 1.) +
                   (g[:, 2] - g[:, 0] + 1.) * (g[:, 3] - g[:, 1] + 1.) - inters)
            overlaps = inters / uni
            ovmax = np.max(overlaps)
            jmax = int(np.argmax(overlaps))
        if ovmax > ovthresh:
            if not hard[jmax]:
                if not taken[i][jmax]:
                    tp[d] = 1.
                    taken[i][jmax] = True
                else:
                    fp[d] = 1.
        else:
            fp[d] = 1.
    fp, tp = np.cumsum(fp), np.cumsum(tp)
    rec = tp / float(npos) if npos > 0 else tp * 0.0
    prec = tp / np.maximum(tp + fp, np.finfo(np.float64).eps)
    return rec, prec, voc_ap(rec, prec, use_07_metric)


def evaluate_detections(all_dets, gt_boxes, gt_labels, gt_difficult=None, num_classes=21, ovthresh=0.5,
                        use_07_metric=True):
    """mAP over classes 1..num_classes-1 (reference scripts/eval_pascal.py:74-96 loop over voc_eval).
    all_dets[i] = detect_image() output of image i; gt_boxes[i] [g,4], gt_labels[i] [g] ints."""
    n = len(all_dets)
    if gt_difficult is None:
        gt_difficult = [np.zeros(len(np.asarray(l).reshape(-1)), dtype=bool) for l in gt_labels]
    aps = []
    for j in range(1, num_classes):
        dets = [all_dets[i][j] for i in range(n)]
        gb, gd = [], []
        for i in range(n):
            lab = np.asarray(gt_labels[i]).reshape(-1)
            sel = lab == j
            gb.append(np.asarray(gt_boxes[i], dtype=np.float64).reshape(-1, 4)[sel])
            gd.append(np.asarray(gt_difficult[i], dtype=bool).reshape(-1)[sel])
        aps.append(voc_eval_arrays(dets, gb, gd, ovthresh, use_07_metric)[2])
    return float(np.mean(aps)), aps

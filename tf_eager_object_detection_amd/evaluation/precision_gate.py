"""Accuracy gate of the float16 throughput mode: the float16 detector against the float32 detector (the reference's
precision) on IDENTICAL weights and images, scored with the reference's own evaluation loop against annotations.

The reference's headline number is two-sided: images/sec AND "mAP delta vs ref" (BASELINE.json).  The float32 detector
is the parity mode (every hot-path stage bit-compared with the restated reference); the float16 detector is the mode
that reaches the throughput target, and what its narrower convolutions do to the DETECTIONS is measured here:

  * annotated synthetic scenes (`labelled_scenes`: coloured rectangles / ellipses on a grey gradient, class = colour),
  * a detector with real signal: there are no checkpoints offline and no backward passes for the fused kernels, but the
    layers that turn features into scores and regressions are linear, so `fit_readout_heads` fits them in closed form
    (ridge regression on the float32 features of the seeded random network, AnchorTarget's / ProposalTarget's labelling
    rules) -- a weak but genuine detector of the scenes' objects (VOC07 mAP ~0.5 on held-out scenes),
  * both precisions run `im_detect` (model/fpn/base_fpn_model.py:364-390) on the same held-out scenes, `detect_image`
    turns both into per-class detections exactly as the reference's evaluation loop does
    (evaluation/pascal_eval_files_utils.py:76-106: score threshold 0.05, per-class NMS 0.3, per-image cap), and both are
    scored against the annotations with the reference's VOC routine (scripts/eval_pascal.py:74-96 over
    detectron_pascal_evaluation_utils.voc_eval): `map_delta` = mAP(float16) - mAP(float32), with a paired bootstrap
    over images for its sampling error,
  * plus what an mAP cannot show: the float32 detections themselves taken as ground truth ("reproduction": every
    differently ranked / placed / thresholded detection counts), agreement of the RPN's kept anchor indices, and the
    |score| / box differences of the detections both modes found.
"""
import numpy as np
import torch

from . import pascal_eval as pe

__all__ = ['labelled_scenes', 'fit_readout_heads', 'detect_batch', 'compare_detections', 'paired_map_delta', 'fp16_vs_fp32']

_MEANS = np.float32([103.939, 116.779, 123.68])


# ---- a detector with real signal: the last layers FITTED on labelled synthetic scenes -------------------------------------
# There are no checkpoints offline and no backward passes for the fused kernels, but the layers that turn features into
# scores and regressions are linear, so they can be fitted in closed form (ridge regression on the float32 features of
# the seeded random network): the RpnHead's two 1x1 convolutions on anchor labels / targets (AnchorTarget's rules:
# model/anchor_target.py), the RoI head's score / bbox layers on proposal labels / targets (ProposalTarget's).  The
# result is a weak but genuine detector of the scenes' objects, which makes "mAP against annotations" measurable for
# both precisions on identical inputs.

def _palette(num_fg):
    """num_fg well separated RGB colours (hue steps, two brightness levels), class c = colour c - 1"""
    import colorsys
    cols = []
    for c in range(num_fg):
        r, g, b = colorsys.hsv_to_rgb((c * 0.618033988749895) % 1.0, 0.9, 1.0 if c % 2 == 0 else 0.62)
        cols.append((255.0 * r, 255.0 * g, 255.0 * b))
    return np.float32(cols)


def labelled_scenes(n, image_shape, seed=0, num_classes=21, noise=6.0, batch=8, device='cuda'):
    """Generator over n scenes with ANNOTATIONS in batches: 3-8 filled rectangles / ellipses that barely overlap, class =
    colour (palette of num_classes - 1 colours, +-12 brightness jitter) on a greyish gradient, pixel noise.  Scene
    parameters come from a seeded numpy generator, the pixels are drawn on the device (seeded torch generator).
    Yields (images [b,H,W,3] float32 mean-subtracted on `device`, gt_boxes [b][g,4] float32 numpy (x1,y1,x2,y2
    inclusive), gt_labels [b][g] int32 numpy)."""
    H, W = int(image_shape[0]), int(image_shape[1])
    rng = np.random.default_rng(seed)
    gen = torch.Generator(device=device)
    gen.manual_seed(int(seed) * 7919 + 13)
    pal = _palette(num_classes - 1)
    yy = torch.arange(H, dtype=torch.float32, device=device).view(H, 1)
    xx = torch.arange(W, dtype=torch.float32, device=device).view(1, W)
    means = torch.tensor(_MEANS, device=device)
    smin, smax = 40.0, 0.55 * min(H, W)
    for s in range(0, n, batch):
        b_n = min(batch, n - s)
        out = torch.empty((b_n, H, W, 3), dtype=torch.float32, device=device)
        gtb, gtl = [], []
        for i in range(b_n):
            a, b = rng.uniform(-1, 1), rng.uniform(-1, 1)
            tint = rng.uniform(-8, 8, 3).astype(np.float32)
            base = 120.0 + 35.0 * (float(a) * (xx / W - 0.5) + float(b) * (yy / H - 0.5))
            img = base.unsqueeze(2) + torch.tensor(tint, device=device).view(1, 1, 3)
            boxes, labels = [], []
            want = int(rng.integers(3, 9))
            for _ in range(200):
                if len(boxes) >= want:
                    break
                sz = np.exp(rng.uniform(np.log(smin), np.log(smax)))
                ar = np.exp(rng.uniform(np.log(0.5), np.log(2.0)))
                w, h = min(sz * np.sqrt(ar), W - 2), min(sz / np.sqrt(ar), H - 2)
                x1, y1 = rng.uniform(0, W - 1 - w), rng.uniform(0, H - 1 - h)
                box = np.float32([np.floor(x1), np.floor(y1), np.floor(x1 + w), np.floor(y1 + h)])
                lab = int(rng.integers(1, num_classes))
                jit = float(rng.uniform(-12, 12))
                ellipse = bool(rng.uniform() < 0.5)
                if boxes and float(_iou_plus1(box.astype(np.float64), np.float64(boxes)).max()) > 0.02:
                    continue
                col = torch.tensor(pal[lab - 1] + jit, device=device).view(1, 1, 3)
                if ellipse:
                    cx, cy = 0.5 * (box[0] + box[2]), 0.5 * (box[1] + box[3])
                    rx, ry = 0.5 * (box[2] - box[0] + 1), 0.5 * (box[3] - box[1] + 1)
                    m = ((xx - float(cx)) / float(rx)) ** 2 + ((yy - float(cy)) / float(ry)) ** 2 <= 1.0
                else:
                    m = (xx >= float(box[0])) & (xx <= float(box[2])) & (yy >= float(box[1])) & (yy <= float(box[3]))
                img = torch.where(m.unsqueeze(2), col, img)
                boxes.append(box)
                labels.append(lab)
            img = img + torch.randn(img.shape, generator=gen, device=device) * noise
            out[i] = img.clamp_(0, 255) - means
            gtb.append(np.float32(boxes).reshape(-1, 4))
            gtl.append(np.int32(labels))
        yield out, gtb, gtl


def _ridge(G, R, lam_rel=1e-3):
    G = G.double()
    d = G.shape[0]
    lam = lam_rel * float(torch.trace(G)) / d
    return torch.linalg.solve(G + lam * torch.eye(d, dtype=torch.float64, device=G.device), R.double())


def _family(model):
    return 'fpn' if hasattr(model, 'l4') else 'frcnn'


def _rpn_anchors(model):
    """[N,4] float32 anchors on the device in the RpnHead's output order (location-major, anchor-minor)"""
    from .. import synthetic as syn
    from ..utils.anchor_generator import make_fpn_anchors, generate_by_anchor_base_tf
    if _family(model) == 'fpn':
        return make_fpn_anchors(model.image_shape, syn.FPN_STRIDES, syn.FPN_BASE_SIZES, syn.FPN_SCALES, syn.FPN_RATIOS)
    hot = model._hot[0]
    return generate_by_anchor_base_tf(hot.anchor_base, hot.stride, hot.fh, hot.fw).contiguous()


def _rpn_activations(model, x):
    """relu(rpn_conv(map)) of every RPN location, [B, locations, cin] float32 in the anchors' location order"""
    from ..model import fpn_detector as fd
    maps = model.features(x)
    maps = maps if isinstance(maps, (list, tuple)) else [maps]
    cin = model.rpn_score.in_channels
    acts = []
    for p in maps:
        a = fd._conv_epi(model.rpn_conv, p if p.is_contiguous(memory_format=torch.channels_last)
                         else p.contiguous(memory_format=torch.channels_last), relu=True)
        acts.append(a.permute(0, 2, 3, 1).reshape(a.shape[0], -1, cin).float())
    return torch.cat(acts, 1)


def _head_inputs(model, x):
    """one pass up to the RoI head: per image (rois [k,4] in the order of the head's rows, head activation [k,d] float32)"""
    B = x.shape[0]
    out = []
    if _family(model) == 'fpn':
        rpn_s, rpn_d, maps = model._dense(x)
        model._hot_to_head(B, rpn_s, rpn_d, maps)
    else:
        model._run_to_head(x)
    model._last_batch = B
    model.check_complete(B)
    for b in range(B):
        hot = model._hot[b]
        k = int(hot.roi_count.item())
        rois = (hot.sorted_rois if _family(model) == 'fpn' else hot.rois)[:k].contiguous()
        out.append((rois, model.head_activation(hot.roi_features[:k]).float()))
    return out


@torch.no_grad()
def fit_readout_heads(model, scenes, rpn_gain=2.0, cls_gain=8.0, pos_iou=0.7, neg_iou=0.3, fg_iou=0.5, ridge=1e-3):
    """Fits model.rpn_score / rpn_bbox / score / bbox (a float32 detector of any family after prepare()) on annotated scenes
    (`scenes`: a callable returning a fresh labelled_scenes generator with batches <= the model's max batch; it is walked
    twice: RPN first, then the RoI head on the fitted RPN's proposals).  Everything before those layers stays the seeded
    random network.  RPN: per anchor shape a class-balanced ridge regression on +-1 labels (AnchorTarget's rules: IoU >=
    pos_iou or the best anchor of an object positive, < neg_iou negative; model/anchor_target.py) and a ridge regression of
    the encoded box targets on the positives; the fg / bg logits are +-rpn_gain times the regressed label, written in the
    family's channel layout (FPN: (bg, fg) interleaved per anchor, base_fpn_model.py:429; Faster R-CNN: [A bg | A fg],
    base_faster_rcnn_model.py:149-152).  RoI head: one-vs-rest ridge classifiers over the proposals' labels (ProposalTarget's
    rule: IoU >= fg_iou with an object -> its class, model/proposal_target.py) scaled by cls_gain, class-agnostic box
    regression replicated per class."""
    from .. import ops
    dev = next(model.parameters()).device
    A = model.A
    fpn = _family(model) == 'fpn'
    anchors = _rpn_anchors(model)
    n = 0
    cin = model.rpn_score.in_channels
    d = cin + 1
    Gp = torch.zeros((A, d, d), dtype=torch.float64, device=dev)
    Gn = torch.zeros((A, d, d), dtype=torch.float64, device=dev)
    sp = torch.zeros((A, d), dtype=torch.float64, device=dev)
    sn = torch.zeros((A, d), dtype=torch.float64, device=dev)
    Bt = torch.zeros((A, d, 4), dtype=torch.float64, device=dev)
    npos = torch.zeros(A, dtype=torch.float64, device=dev)
    nneg = torch.zeros(A, dtype=torch.float64, device=dev)
    for x, gt_boxes, gt_labels in scenes():
        n += int(x.shape[0])
        F_ = _rpn_activations(model, x)                                    # [B, locations, cin], location-major like the anchors
        ones = torch.ones(F_.shape[1], 1, device=dev)
        for b in range(F_.shape[0]):
            gt = torch.from_numpy(gt_boxes[b]).to(dev)
            iou = ops.pairwise_iou(anchors, gt)                            # [N, g]
            best, arg = iou.max(dim=1)
            pos = best >= pos_iou
            pos[iou.argmax(dim=0)] = True                                  # the best anchor of every object (AnchorTarget)
            neg = (best < neg_iou) & ~pos
            tgt = ops.encode(anchors, gt[arg].contiguous(), [0, 0, 0, 0], [1, 1, 1, 1])
            X = torch.cat([F_[b], ones], 1)                                # [locations, cin + 1]
            pos, neg, tgt = pos.view(-1, A), neg.view(-1, A), tgt.view(-1, A, 4)
            for a in range(A):
                Xp, Xn = X[pos[:, a]], X[neg[:, a]]
                Gp[a] += (Xp.t() @ Xp).double()
                Gn[a] += (Xn.t() @ Xn).double()
                sp[a] += Xp.sum(0).double()
                sn[a] += Xn.sum(0).double()
                Bt[a] += (Xp.t() @ tgt[pos[:, a], a]).double()
                npos[a] += Xp.shape[0]
                nneg[a] += Xn.shape[0]
    ws, bs, wb, bb = model.rpn_score.weight, model.rpn_score.bias, model.rpn_bbox.weight, model.rpn_bbox.bias
    for a in range(A):
        if float(npos[a]) < 1 or float(nneg[a]) < 1:
            continue                                                       # (an anchor shape no object ever matched: stays random)
        th = _ridge(Gp[a] / npos[a] + Gn[a] / nneg[a], (sp[a] / npos[a] - sn[a] / nneg[a]).unsqueeze(1), ridge)[:, 0].float()
        i_fg, i_bg = (2 * a + 1, 2 * a) if fpn else (A + a, a)
        ws[i_fg, :, 0, 0], bs[i_fg] = rpn_gain * th[:cin], rpn_gain * th[cin]
        ws[i_bg, :, 0, 0], bs[i_bg] = -rpn_gain * th[:cin], -rpn_gain * th[cin]
        tb = _ridge(Gp[a] / npos[a], Bt[a] / npos[a], max(ridge, 1e-2)).float()        # [cin + 1, 4]
        for k in range(4):
            wb[4 * a + k, :, 0, 0], bb[4 * a + k] = tb[:cin, k], tb[cin, k]
    model._rpn_pair = None
    # ---- RoI head on the proposals of the fitted RPN (ProposalTarget's labels: IoU >= 0.5 with an object -> its class)
    ncls = model.num_classes
    dh = model.score.in_features + 1
    Gc = torch.zeros((ncls, dh, dh), dtype=torch.float64, device=dev)
    sc = torch.zeros((ncls, dh), dtype=torch.float64, device=dev)
    nc = torch.zeros(ncls, dtype=torch.float64, device=dev)
    Gf = torch.zeros((dh, dh), dtype=torch.float64, device=dev)
    Bf = torch.zeros((dh, 4), dtype=torch.float64, device=dev)
    nfg = 0
    stds = list(model._hot[0].cfg['roi_stds'])
    for x, gt_boxes, gt_labels in scenes():
        for b, (rois, h) in enumerate(_head_inputs(model, x)):
            k = rois.shape[0]
            X = torch.cat([h, torch.ones(k, 1, device=dev)], 1)
            gt = torch.from_numpy(gt_boxes[b]).to(dev)
            gl = torch.from_numpy(gt_labels[b].astype(np.int64)).to(dev)
            iou = ops.pairwise_iou(rois, gt)
            best, arg = iou.max(dim=1)
            lab = torch.where(best >= fg_iou, gl[arg], torch.zeros_like(arg))
            for c in torch.unique(lab).tolist():
                Xc = X[lab == c]
                Gc[c] += (Xc.t() @ Xc).double()
                sc[c] += Xc.sum(0).double()
                nc[c] += Xc.shape[0]
            fg = lab > 0
            if bool(fg.any()):
                Xf = X[fg]
                t = ops.encode(rois[fg].contiguous(), gt[arg[fg]].contiguous(), [0, 0, 0, 0], stds)
                Gf += (Xf.t() @ Xf).double()
                Bf += (Xf.t() @ t).double()
                nfg += int(fg.sum())
    have = nc > 0
    G = (Gc[have] / nc[have].view(-1, 1, 1)).sum(0)
    R = torch.zeros((dh, ncls), dtype=torch.float64, device=dev)
    for c in range(ncls):
        if bool(have[c]):
            R[:, c] = sc[c] / nc[c]
    th = _ridge(G, R, ridge).float()                                              # [1025, ncls]: one-hot targets, class-balanced
    model.score.weight.copy_(cls_gain * th[:-1].t())
    model.score.bias.copy_(cls_gain * th[-1])
    model.score.bias[~have] = -1e4                                         # (a class the fit never saw is never predicted)
    tb = _ridge(Gf / max(nfg, 1), Bf / max(nfg, 1), max(ridge, 1e-2)).float()          # [1025, 4], class-agnostic, replicated
    model.bbox.weight.copy_(tb[:-1].t().repeat(ncls, 1))
    model.bbox.bias.copy_(tb[-1].repeat(ncls))
    return dict(train_images=int(n), rpn_pos_anchors=int(npos.sum()), rpn_neg_anchors=int(nneg.sum()),
                roi_fg=int(nfg), roi_per_class=[int(v) for v in nc.tolist()])


@torch.no_grad()
def detect_batch(model, x, score_threshold=0.05, iou_threshold=0.3, max_per_class=50, max_per_image=50):
    """x [b,H,W,3] on the device, b <= the model's max batch.  -> per image (detections per class as detect_image
    returns them, kept anchor indices of the RPN NMS as a sorted numpy array)."""
    H, W = model.image_shape
    dets, kept = [], []
    for b, (scores, deltas, rois) in enumerate(model.im_detect(x, 1.0)):
        dets.append(pe.detect_image(scores, deltas, rois, 1.0, H, W, num_classes=model.num_classes,
                                    score_threshold=score_threshold, iou_threshold=iou_threshold,
                                    max_objects_per_class=max_per_class, max_objects_per_image=max_per_image))
        hot = model._hot[b]
        kept.append(np.sort(hot.roi_idx[:int(hot.roi_count.item())].cpu().numpy()))
    return dets, kept


def _iou_plus1(b, g):
    iw = np.maximum(np.minimum(g[:, 2], b[2]) - np.maximum(g[:, 0], b[0]) + 1., 0.)
    ih = np.maximum(np.minimum(g[:, 3], b[3]) - np.maximum(g[:, 1], b[1]) + 1., 0.)
    inter = iw * ih
    return inter / ((b[2] - b[0] + 1.) * (b[3] - b[1] + 1.) + (g[:, 2] - g[:, 0] + 1.) * (g[:, 3] - g[:, 1] + 1.) - inter)


def compare_detections(dets_ref, dets_test, num_classes=21, gt_score_floor=0.0, use_07_metric=True, dets_gt=None):
    """Ground truth = the detections `dets_gt` (default: dets_ref themselves) with score >= gt_score_floor; dets_ref and
    dets_test are both scored against it."""
    gb, gl = [], []
    for d in (dets_ref if dets_gt is None else dets_gt):
        boxes = [d[j][d[j][:, 4] >= gt_score_floor, :4] for j in range(1, num_classes)]
        labels = [np.full(len(b), j, np.int32) for j, b in zip(range(1, num_classes), boxes)]
        gb.append(np.concatenate(boxes, 0) if boxes else np.zeros((0, 4), np.float32))
        gl.append(np.concatenate(labels, 0) if labels else np.zeros(0, np.int32))
    # classes without any ground-truth box would score AP 0 for both sides; they are left out of the mean
    present = sorted(set(int(l) for g in gl for l in g))
    m_ref, ap_ref = pe.evaluate_detections(dets_ref, gb, gl, num_classes=num_classes, use_07_metric=use_07_metric)
    m_test, ap_test = pe.evaluate_detections(dets_test, gb, gl, num_classes=num_classes, use_07_metric=use_07_metric)
    sel = [j - 1 for j in present]
    m_ref = float(np.mean([ap_ref[i] for i in sel])) if sel else 0.0
    m_test = float(np.mean([ap_test[i] for i in sel])) if sel else 0.0
    # detections both modes found: same class, IoU(+1) > 0.5 with the best partner
    dscore, dbox, matched, total = [], [], 0, 0
    for dr, dt in zip(dets_ref, dets_test):
        for j in range(1, num_classes):
            r, t = dr[j], dt[j]
            total += len(r)
            if len(r) == 0 or len(t) == 0:
                continue
            for row in r:
                iou = _iou_plus1(row[:4].astype(np.float64), t[:, :4].astype(np.float64))
                k = int(np.argmax(iou))
                if iou[k] > 0.5:
                    matched += 1
                    dscore.append(abs(float(row[4]) - float(t[k, 4])))
                    dbox.append(float(np.abs(row[:4] - t[k, :4]).max()))
    return dict(map_ref=m_ref, map_test=m_test, map_delta=m_test - m_ref, classes_scored=len(sel),
                gt_boxes=int(sum(len(g) for g in gl)), ref_detections=int(total),
                test_detections=int(sum(len(d[j]) for d in dets_test for j in range(1, num_classes))),
                matched_fraction=(matched / total) if total else 1.0,
                max_abs_dscore=float(np.max(dscore)) if dscore else 0.0,
                median_abs_dscore=float(np.median(dscore)) if dscore else 0.0,
                p99_abs_dscore=float(np.percentile(dscore, 99)) if dscore else 0.0,
                max_abs_dbox_px=float(np.max(dbox)) if dbox else 0.0,
                median_abs_dbox_px=float(np.median(dbox)) if dbox else 0.0)


def _image_matches(dets, gt_boxes, gt_labels, num_classes, ovthresh=0.5):
    """Per class and image: (scores desc, tp flags, number of ground-truth boxes) -- voc_eval's matching only depends on
    the order of an image's own detections, so it can be done once per image and re-used by every bootstrap resample."""
    out = []
    for j in range(1, num_classes):
        per = []
        for d, gb, gl in zip(dets, gt_boxes, gt_labels):
            g = np.asarray(gb, np.float64).reshape(-1, 4)[np.asarray(gl).reshape(-1) == j]
            dj = np.asarray(d[j], np.float64).reshape(-1, 5)
            order = np.argsort(-dj[:, 4], kind='stable')
            dj = dj[order]
            tp = np.zeros(len(dj), bool)
            taken = np.zeros(len(g), bool)
            for k in range(len(dj)):
                if len(g) == 0:
                    continue
                ov = _iou_plus1(dj[k, :4], g)
                m = int(np.argmax(ov))
                if ov[m] > ovthresh and not taken[m]:
                    tp[k] = True
                    taken[m] = True
            per.append((dj[:, 4], tp, len(g)))
        out.append(per)
    return out


def _map_from_matches(matches, idx, use_07_metric=True):
    """mAP over the classes that have ground truth among images `idx` (the others are left out of the mean)"""
    aps = []
    for per in matches:
        npos = sum(per[i][2] for i in idx)
        if npos == 0:
            continue
        sc = np.concatenate([per[i][0] for i in idx])
        tp = np.concatenate([per[i][1] for i in idx])
        order = np.argsort(-sc, kind='stable')
        tp = tp[order]
        ctp, cfp = np.cumsum(tp), np.cumsum(~tp)
        rec = ctp / float(npos)
        prec = ctp / np.maximum(ctp + cfp, np.finfo(np.float64).eps)
        aps.append(pe.voc_ap(rec, prec, use_07_metric))
    return float(np.mean(aps)) if aps else 0.0


def _flat_matches(matches):
    """per class: (scores, tp flags, image index) of all images' detections sorted by score (stable), ground-truth boxes per
    image -- what a resample needs to re-weight"""
    out = []
    for per in matches:
        sc = np.concatenate([p_[0] for p_ in per]) if per else np.zeros(0)
        tp = np.concatenate([p_[1] for p_ in per]) if per else np.zeros(0, bool)
        im = np.concatenate([np.full(len(p_[0]), i, np.int64) for i, p_ in enumerate(per)]) if per else np.zeros(0, np.int64)
        order = np.argsort(-sc, kind='stable')
        out.append((tp[order], im[order], np.asarray([p_[2] for p_ in per], np.float64)))
    return out


def _map_weighted(flat, counts, use_07_metric=True):
    """mAP of the image multiset `counts` (image i drawn counts[i] times): every detection of image i counts counts[i] times.
    Equal to _map_from_matches on the expanded index list -- the copies of a detection are adjacent in the score order and
    only the last copy of a true positive / the point before the first copy of a false positive can be a maximum of the
    precision envelope -- in O(detections) per class instead of a Python loop over images."""
    aps = []
    for tp, im, npos_img in flat:
        npos = float(np.dot(counts, npos_img))
        if npos == 0:
            continue
        w = counts[im].astype(np.float64)
        ctp, cfp = np.cumsum(tp * w), np.cumsum((~tp) * w)
        rec = ctp / npos
        prec = ctp / np.maximum(ctp + cfp, np.finfo(np.float64).eps)
        aps.append(pe.voc_ap(rec, prec, use_07_metric))
    return float(np.mean(aps)) if aps else 0.0


def paired_map_delta(dets_a, dets_b, gt_boxes, gt_labels, num_classes=21, resamples=400, seed=0, use_07_metric=True):
    """mAP of two detection sets against the same annotations, their difference, and a paired bootstrap over images
    (the same resampled image set for both) of that difference."""
    ma = _image_matches(dets_a, gt_boxes, gt_labels, num_classes)
    mb = _image_matches(dets_b, gt_boxes, gt_labels, num_classes)
    n = len(dets_a)
    full = np.arange(n)
    a, b = _map_from_matches(ma, full, use_07_metric), _map_from_matches(mb, full, use_07_metric)
    fa, fb = _flat_matches(ma), _flat_matches(mb)
    rng = np.random.default_rng(seed)
    ds = []
    for _ in range(resamples):
        counts = np.bincount(rng.integers(0, n, n), minlength=n)
        ds.append(_map_weighted(fb, counts, use_07_metric) - _map_weighted(fa, counts, use_07_metric))
    ds = np.asarray(ds) if ds else np.zeros(1)
    return dict(map_a=a, map_b=b, delta=b - a, delta_boot_mean=float(ds.mean()), delta_boot_std=float(ds.std()),
                delta_ci95=[float(np.percentile(ds, 2.5)), float(np.percentile(ds, 97.5))], resamples=resamples)


_FAMILIES = {'fpn': ('ResNet-%d-FPN', (800, 1333), 1000), 'c4': ('ResNet-%d C4 Faster R-CNN', (800, 1333), 300),
             'vgg16': ('VGG16 Faster R-CNN', (600, 800), 300)}


def _build(family, depth, num_classes, image_shape, num_proposals, dtype, max_batch, hot_kwargs):
    from ..model.fpn_detector import ResNetFpnDetector
    from ..model.frcnn_detector import ResNetC4Detector, Vgg16Detector
    if family == 'fpn':
        return ResNetFpnDetector(depth, num_classes, image_shape, num_proposals, dtype=dtype, max_batch=max_batch, **hot_kwargs)
    if family == 'c4':
        return ResNetC4Detector(depth, num_classes, image_shape, num_proposals, dtype=dtype, max_batch=max_batch, **hot_kwargs)
    return Vgg16Detector(num_classes, image_shape, num_proposals, dtype=dtype, max_batch=max_batch, **hot_kwargs)


def fp16_vs_fp32(num_images=256, image_shape=None, depth=None, num_classes=21, num_proposals=None, batch32=4,
                 batch16=8, seed=0, train_images=64, ridge=1e-3, resamples=400, family='fpn', test_mode='fp16', **hot_kwargs):
    """The whole gate for one detector family ('fpn': ResNet-101-FPN @ 800x1333, 'c4': ResNet-50 C4 @ 800x1333, 'vgg16':
    VGG16 @ 600x800 -- BASELINE configs 3 / 2 / 1).  -> dict for bench.py's `e2e.*.map_delta_vs_fp32` and the GPU tests;
    `within_bar` = the point estimate is inside the north star's +-0.002; `ci_half_width_within_bar` = the paired-bootstrap
    95 % interval is narrower than the bar on both sides of its mean (the gate can tell +-0.002 from noise at this number of
    scenes -- a statement about the gate's RESOLUTION, not about the mode); `ci_inside_bar` = BOTH ends of that interval lie
    within +-0.002 (the statement about the mode: only this one supports "mAP within +-0.002").
    test_mode: 'fp16' (the float16 throughput mode) or 'x3' / 'x2' (the float32 split-precision modes, csrc/conv_x3.hip) as the
    detector under test; the reference detector is always the exact-float32 mode (the `*_fp16` keys then hold the x3 figures)."""
    name, shape0, prop0 = _FAMILIES[family]
    image_shape = tuple(shape0 if image_shape is None else image_shape)
    num_proposals = prop0 if num_proposals is None else num_proposals
    depth = (101 if family == 'fpn' else 50) if depth is None else depth
    torch.manual_seed(seed)
    m32 = _build(family, depth, num_classes, image_shape, num_proposals, torch.float32, batch32, hot_kwargs).prepare()
    fit = fit_readout_heads(m32, lambda: labelled_scenes(train_images, image_shape, seed=seed + 5, num_classes=num_classes,
                                                         batch=batch32), ridge=ridge)
    fit['ridge'] = ridge
    state = {k: v.detach().clone() for k, v in m32.state_dict().items()}
    if test_mode in ('x3', 'x2'):
        m16 = _build(family, depth, num_classes, image_shape, num_proposals, torch.float32, batch16, dict(hot_kwargs, f32_form=test_mode))
    else:
        m16 = _build(family, depth, num_classes, image_shape, num_proposals, torch.float16, batch16, hot_kwargs)
    m16.load_state_dict(state)
    m16.prepare()
    d32, k32, d16, k16, gtb, gtl = [], [], [], [], [], []
    for x, gb, gl in labelled_scenes(num_images, image_shape, seed=seed + 17, num_classes=num_classes, batch=batch16):
        for s in range(0, x.shape[0], batch32):
            d, k = detect_batch(m32, x[s:s + batch32])
            d32 += d
            k32 += k
        d, k = detect_batch(m16, x)
        d16 += d
        k16 += k
        gtb += gb
        gtl += gl
    pair = paired_map_delta(d32, d16, gtb, gtl, num_classes, resamples=resamples, seed=seed)
    pair_area = paired_map_delta(d32, d16, gtb, gtl, num_classes, resamples=0, seed=seed, use_07_metric=False)
    repro = compare_detections(d32, d16, num_classes, 0.0)
    agree = [len(np.intersect1d(a, b, assume_unique=True)) / max(len(a), 1) for a, b in zip(k32, k16)]
    lo, hi = pair['delta_ci95']
    rec = dict(images=num_images, image=list(image_shape), model=(name % depth) if '%d' in name else name, family=family,
               metric='VOC07 11-point mAP',
               test_mode={'fp16': 'float16 throughput mode', 'x3': 'float32 split-precision mode (three bfloat16 limbs, six products '
                          'per k, float32 accumulation); the *_fp16 keys hold ITS figures',
                          'x2': 'float32 split-precision mode, two float16 limbs (three products per k, float32 accumulation, '
                          'float16 range); the *_fp16 keys hold ITS figures'}[test_mode],
               protocol='annotated synthetic scenes (coloured rectangles / ellipses, class = colour); float32 (parity mode) '
                        'and float16 detector with the SAME weights on the SAME images: im_detect -> detect_image (score >= '
                        '0.05, per-class NMS 0.3, 50 per image; evaluation/pascal_eval_files_utils.py:76-106) -> VOC07 mAP '
                        'against the annotations (scripts/eval_pascal.py:74-96); map_delta = mAP(fp16) - mAP(fp32)',
               weights='seeded random-init backbone / neck / RPN conv / FC layers; the last linear layers of the RPN and RoI '
                       'heads fitted in closed form (ridge regression on float32 features) on %d other annotated scenes'
                       % train_images,
               data='synthetic', map_fp32=pair['map_a'], map_fp16=pair['map_b'], map_delta=pair['delta'],
               map_delta_ci95_paired_bootstrap=pair['delta_ci95'], map_delta_bootstrap_std=pair['delta_boot_std'],
               bar=0.002, within_bar=bool(abs(pair['delta']) <= 0.002),
               ci_half_width_within_bar=bool(max(abs(lo - pair['delta_boot_mean']), abs(hi - pair['delta_boot_mean'])) <= 0.002),
               ci_inside_bar=bool(lo >= -0.002 and hi <= 0.002),
               map_delta_area_metric=pair_area['delta'], gt_boxes=int(sum(len(g) for g in gtl)),
               classes_scored=len(set(int(l) for g in gtl for l in g)),
               reproduction={'protocol': 'the float32 detections themselves as ground truth (mAP fp32 = 1 by construction): '
                                         'every differently ranked / placed / thresholded detection counts',
                             'map_fp16': repro['map_test'], 'map_delta_reproduction': repro['map_delta'],
                             'matched_fraction': repro['matched_fraction']},
               detections_fp32=repro['ref_detections'], detections_fp16=repro['test_detections'],
               max_abs_dscore=repro['max_abs_dscore'], p99_abs_dscore=repro['p99_abs_dscore'],
               median_abs_dscore=repro['median_abs_dscore'], max_abs_dbox_px=repro['max_abs_dbox_px'],
               median_abs_dbox_px=repro['median_abs_dbox_px'],
               rpn_kept_index_agreement_mean=float(np.mean(agree)), rpn_kept_index_agreement_min=float(np.min(agree)),
               fit=fit)
    del m32, m16
    torch.cuda.empty_cache()
    return rec

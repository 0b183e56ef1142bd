"""AnchorTarget / ProposalTarget / losses (reference model/anchor_target.py,
model/proposal_target.py, model/losses.py).  Exact parity on the deterministic parts, distribution-level checks
after the random sub-sampling (the reference samples with tf.random_shuffle / np.random.choice)."""
import numpy as np
import pytest
import torch

from oracle import c_oracle as co
from oracle import oracle_np as on
from tf_eager_object_detection_amd import synthetic as syn


def test_losses_cpu():
    from tf_eager_object_detection_amd.model.losses import cls_loss, smooth_l1_loss
    rng = np.random.default_rng(0)
    logits = rng.standard_normal((50, 5)).astype(np.float32)
    labels = rng.integers(0, 5, 50)
    p = np.exp(logits - logits.max(1, keepdims=True)); p /= p.sum(1, keepdims=True)
    want = float(np.mean(-np.log(p[np.arange(50), labels])))
    assert abs(float(cls_loss(torch.from_numpy(logits), torch.from_numpy(labels))) - want) < 1e-5
    w = (rng.random(50) < 0.5).astype(np.float32)
    want_w = float(np.sum(-np.log(p[np.arange(50), labels]) * w) / max(1, int((w != 0).sum())))
    assert abs(float(cls_loss(torch.from_numpy(logits), torch.from_numpy(labels), torch.from_numpy(w))) - want_w) < 1e-5
    pred, tgt = rng.standard_normal((20, 8)).astype(np.float32), rng.standard_normal((20, 8)).astype(np.float32)
    iw, ow = (rng.random((20, 8)) < 0.5).astype(np.float32), rng.random((20, 8)).astype(np.float32)
    for sigma in (1.0, 3.0):
        s2 = sigma ** 2
        d = iw * (pred - tgt)
        a = np.abs(d)
        sign = (a < 1.0 / s2).astype(np.float32)
        il = d ** 2 * (s2 / 2.0) * sign + (a - 0.5 / s2) * (1.0 - sign)
        want = float(np.mean(np.sum(ow * il, axis=1)))
        got = float(smooth_l1_loss(torch.from_numpy(pred), torch.from_numpy(tgt), torch.from_numpy(iw), torch.from_numpy(ow), sigma))
        assert abs(got - want) < 1e-5


@pytest.mark.gpu
def test_anchor_target_matches_oracle_and_sampling_limits():
    from tf_eager_object_detection_amd.model.anchor_target import AnchorTarget
    rng = np.random.default_rng(11)
    shape = (320, 480)
    anchors = co.fpn_anchors(shape)
    gt = syn.random_boxes(7, shape, rng, 30, 200)
    gen = torch.Generator(device='cuda'); gen.manual_seed(3)
    at = AnchorTarget(0.7, 0.3, 256, 128, [0, 0, 0, 0], [1, 1, 1, 1], generator=gen)
    ga, gg = torch.from_numpy(anchors).cuda(), torch.from_numpy(gt).cuda()
    idx, an, labels0, argmax = at.labels_before_sampling(gg, shape, ga)
    w_idx, w_labels, w_argmax = on.anchor_target_labels(gt, shape, anchors, 0.7, 0.3)
    np.testing.assert_array_equal(idx.cpu().numpy(), w_idx)
    np.testing.assert_array_equal(labels0.cpu().numpy(), w_labels)
    np.testing.assert_array_equal(argmax.cpu().numpy(), w_argmax)
    labels, targets, inside, outside = at((gg, shape, ga))
    labels, targets, inside, outside = (t.cpu().numpy() for t in (labels, targets, inside, outside))
    n = anchors.shape[0]
    assert labels.shape == (n,) and targets.shape == (n, 4) and inside.shape == (n, 4) and outside.shape == (n, 4)
    full0 = -np.ones(n, np.float32); full0[w_idx] = w_labels
    assert set(np.unique(labels)) <= {-1.0, 0.0, 1.0}
    assert np.all(full0[labels == 1] == 1) and np.all(full0[labels == 0] == 0)        # sampling only disables
    nfg, nbg = int((labels == 1).sum()), int((labels == 0).sum())
    assert nfg == min(128, int((w_labels == 1).sum())) and nfg + nbg == min(256, nfg + int((w_labels == 0).sum()))
    enc = on.encode_bbox_with_mean_and_std(anchors[w_idx], gt[w_argmax], [0, 0, 0, 0], [1, 1, 1, 1])
    np.testing.assert_allclose(targets[w_idx], enc, rtol=1e-5, atol=1e-5)
    outside_idx = np.setdiff1d(np.arange(n), w_idx)
    assert np.all(targets[outside_idx] == 0) and np.all(labels[outside_idx] == -1)
    assert np.all(inside[labels == 1] == 1) and np.all(inside[labels != 1] == 0)
    np.testing.assert_allclose(outside[labels >= 0], 1.0 / (nfg + nbg), rtol=1e-6)
    assert np.all(outside[labels < 0] == 0)


@pytest.mark.gpu
def test_proposal_target_matches_oracle_and_sampling_limits():
    from tf_eager_object_detection_amd.model.proposal_target import ProposalTarget
    rng = np.random.default_rng(12)
    shape = (600, 800)
    gt = syn.random_boxes(6, shape, rng, 60, 300)
    gt_labels = rng.integers(1, 21, 6).astype(np.int64)
    rois = np.concatenate([syn.random_boxes(400, shape, rng, 20, 300),
                           (gt[rng.integers(0, 6, 200)] + rng.normal(0, 8, (200, 4))).astype(np.float32), gt]).astype(np.float32)
    gr, gg, gl = torch.from_numpy(rois).cuda(), torch.from_numpy(gt).cuda(), torch.from_numpy(gt_labels).cuda()
    stds = [0.1, 0.1, 0.2, 0.2]
    for quirk, neg in ((True, 0.1), (False, 0.0)):     # (0.1: fewer background RoIs than wanted -> sampled with replacement)
        gen = torch.Generator(device='cuda'); gen.manual_seed(5)
        pt = ProposalTarget(21, 0.5, neg, 128, 32, [0, 0, 0, 0], stds, generator=gen, reference_row_labels=quirk)
        labels, ga, fg, bg = pt.assign(gr, gg, gl)
        w_labels, w_ga, w_fg, w_bg = on.proposal_target_assign(rois, gt, gt_labels, 0.5, neg)
        np.testing.assert_array_equal(labels.cpu().numpy(), w_labels)
        np.testing.assert_array_equal(ga.cpu().numpy(), w_ga)
        np.testing.assert_array_equal(fg.cpu().numpy(), w_fg)
        np.testing.assert_array_equal(bg.cpu().numpy(), w_bg)
        assert len(w_fg) > 32 and ((len(w_bg) < 96) if neg > 0 else (len(w_bg) > 96))
        f_rois, f_labels, f_targets, inside, outside = (t.cpu().numpy() for t in pt((gr, gg, gl)))
        assert f_rois.shape == (128, 4) and f_labels.shape == (128,) and f_targets.shape == (128, 84)
        assert np.all(f_labels[32:] == 0) and np.all(f_labels[:32] > 0) and np.all(outside == 1)
        # every sampled row is one of the input RoIs of the right kind
        def rows_in(a, b):
            return all(any(np.array_equal(r, q) for q in b) for r in a)
        assert rows_in(f_rois[:32], rois[w_fg]) and rows_in(f_rois[32:], rois[w_bg])
        ins = inside.reshape(128, 21, 4)
        tg = f_targets.reshape(128, 21, 4)
        assert np.all(ins[32:] == 0) and np.all(tg[32:] == 0)
        for r in range(32):
            src = int(np.nonzero(np.all(rois == f_rois[r], axis=1))[0][0])
            col = int(w_labels[r]) if quirk else int(w_labels[src])
            assert np.all(ins[r, col] == 1) and ins[r].sum() == 4
            enc = on.encode_bbox_with_mean_and_std(f_rois[r:r + 1], gt[w_ga[src]:w_ga[src] + 1], [0, 0, 0, 0], stds)[0]
            np.testing.assert_allclose(tg[r, col], enc, rtol=1e-4, atol=1e-4)

"""GPU parity tests (run with -m gpu on an MI355X): every HIP kernel, called through the C ABI
(ctypes -> libodet_hip.so), against the CPU oracle on the same seeded inputs.

Bars (BASELINE.json north_star): anchor / kept-box / filter INDICES bit-exact; box coordinates,
scores and RoI features within 1e-4 (fp32).  The IEEE-only kernels (anchors, clip, bilinear
crops, max-pool) are additionally required to be bit-identical.
"""
import numpy as np
import pytest
import torch

from oracle import c_oracle as co
from oracle import oracle_np as on
from tf_eager_object_detection_amd import ops
from tf_eager_object_detection_amd import synthetic as syn

pytestmark = pytest.mark.gpu

TOL = 1e-4          # north_star: bbox coordinates and scores within 1e-4 fp32
M0 = [0, 0, 0, 0]
S1 = [1, 1, 1, 1]
S2 = [0.1, 0.1, 0.2, 0.2]


def g(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def h(t):
    return t.detach().cpu().numpy()


def ulp_distance(a, b):
    """element-wise distance of two float32 arrays in units in the last place (0 = identical bits; +0 == -0)"""
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    ia = a.view(np.int32).astype(np.int64)
    ib = b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, np.int64(-2 ** 31) - ia, ia)        # order-preserving map of the sign-magnitude encoding
    ib = np.where(ib < 0, np.int64(-2 ** 31) - ib, ib)
    return np.abs(ia - ib)


ULP_SEEN = {}       # test-name -> histogram of the ulp distances it saw (printed by the last test of this file)


def close(a, b, max_ulp=0, what=''):
    """The parity bar.  max_ulp = 0: identical bits (every output whose arithmetic is IEEE +, -, *, /, sqrt, floor,
    min / max in the reference's order: anchors, clip, IoU, bilinear crops, pools, gathered scores).  max_ulp = 1: values
    that went through exp / log (box decode / encode, softmax): both sides round a float64 exp / log to float32, the
    float64 libraries (glibc, ROCm's ocml) may differ in their last bit, which moves the float32 result by at most
    one ulp (Eigen's pexp is 1-2 ulp from either).  Far inside north_star's 1e-4."""
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    if not a.size:
        return
    assert np.array_equal(np.isnan(a), np.isnan(b))
    d = ulp_distance(np.nan_to_num(a), np.nan_to_num(b))
    hist = np.bincount(np.minimum(d, 4).reshape(-1), minlength=5)
    key = what or 'unnamed'
    ULP_SEEN[key] = ULP_SEEN.get(key, np.zeros(5, np.int64)) + hist
    worst = int(d.max())
    assert worst <= max_ulp, '%s: max distance %d ulp (allowed %d); histogram 0/1/2/3/4+ ulp: %s; max abs diff %g' % (
        key, worst, max_ulp, hist.tolist(), float(np.max(np.abs(a.astype(np.float64) - b.astype(np.float64)))))


# ------------------------------------------------------------------------------- anchors ------
def test_anchors_shift_bit_exact():
    from tf_eager_object_detection_amd.utils.anchor_generator import generate_anchor_base, generate_by_anchor_base_tf
    for scales, (fh, fw) in (((8, 16, 32), (38, 50)), ((8, 16, 32), (50, 84)), ((4, 8, 16, 32), (2, 3))):
        base = generate_anchor_base(16, [0.5, 1, 2], np.array(scales)).astype(np.float32)
        got = h(generate_by_anchor_base_tf(base, 16, fh, fw))
        np.testing.assert_array_equal(got, on.generate_by_anchor_base_tf(base, 16, fh, fw))
    assert h(generate_by_anchor_base_tf(base, 16, 50, 84)).shape[0] == 50 * 84 * 12


@pytest.mark.parametrize('shape', [(800, 1333), (1333, 1333), (600, 800), (37, 53)])
def test_fpn_anchors_bit_exact(shape):
    from tf_eager_object_detection_amd.utils.anchor_generator import make_anchors, make_fpn_anchors
    got = h(make_fpn_anchors(shape, syn.FPN_STRIDES, syn.FPN_BASE_SIZES, syn.FPN_SCALES, syn.FPN_RATIOS))
    want = co.fpn_anchors(shape)
    assert got.shape[0] == syn.num_fpn_anchors(shape)
    np.testing.assert_array_equal(got, want)
    one = h(make_anchors(64, (1., 2.), (0.5, 1.0, 2.0), 4.0, 5.0, 8))
    np.testing.assert_array_equal(one, on.make_anchors(64, (1., 2.), (0.5, 1.0, 2.0), 4.0, 5.0, 8))


# ------------------------------------------------------------------------- box transforms -----
def test_decode_encode_clip():
    from tf_eager_object_detection_amd.utils.bbox_transform import (decode_bbox_with_mean_and_std,
                                                                    encode_bbox_with_mean_and_std)
    rng = np.random.default_rng(11)
    anchors = co.fpn_anchors((600, 800))
    n = anchors.shape[0]
    for sigma, stds in ((0.1, S1), (0.01, S1), (1.0, S2)):
        d = syn.rpn_deltas(n, rng, sigma) if sigma < 1 else rng.normal(0, 1, (n, 4)).astype(np.float32)
        got = h(decode_bbox_with_mean_and_std(g(anchors), g(d), M0, stds))
        want = co.decode(anchors, d, M0, stds)
        close(got, want, 1, 'decode')
        assert np.mean(got == want) > 0.9999          # correctly rounded exp on both sides
    gt = anchors + rng.uniform(-3, 3, anchors.shape).astype(np.float32)
    gt[:, 2:] = np.maximum(gt[:, 2:], gt[:, :2] + 1)
    close(h(encode_bbox_with_mean_and_std(g(anchors), g(gt), M0, S2)), co.encode(anchors, gt, M0, S2), 1, 'encode')
    # KA5
    z = h(decode_bbox_with_mean_and_std(g(anchors[:100]), g(np.zeros((100, 4), np.float32)), M0, S1))
    np.testing.assert_allclose(z, anchors[:100] + np.float32([0, 0, 1, 1]), rtol=0, atol=1e-5)
    np.testing.assert_array_equal(z, co.decode(anchors[:100], np.zeros((100, 4), np.float32), M0, S1))
    # fused decode+clip == decode then clip
    d = syn.rpn_deltas(n, rng, 0.3)
    fused = h(ops.decode(g(anchors), g(d), M0, S1, clip_shape=(600, 800)))
    want, _ = co.clip_filter(co.decode(anchors, d, M0, S1), 0, 600, 800)
    close(fused, want, 1, 'decode+clip')
    assert fused.min() >= 0 and fused[:, 0::2].max() <= 799 and fused[:, 1::2].max() <= 599


def test_filters_index_exact():
    from tf_eager_object_detection_amd.utils.bbox_tf import bboxes_clip_filter, bboxes_range_filter
    rng = np.random.default_rng(12)
    anchors = co.fpn_anchors((600, 800))
    boxes = co.decode(anchors, syn.rpn_deltas(anchors.shape[0], rng, 0.2), M0, S1)
    gb, gi = bboxes_clip_filter(g(boxes), 0, 600, 800, 16)
    wb, wi = co.clip_filter(boxes, 0, 600, 800, 16)
    assert gi.dtype == torch.int64
    np.testing.assert_array_equal(h(gi), wi)
    np.testing.assert_array_equal(h(gb), wb)
    cb, ci = bboxes_clip_filter(g(boxes), 0, 600, 800)
    np.testing.assert_array_equal(h(cb), co.clip_filter(boxes, 0, 600, 800)[0])
    np.testing.assert_array_equal(h(ci), np.arange(boxes.shape[0]))
    np.testing.assert_array_equal(h(bboxes_range_filter(g(anchors), 600, 800)), co.range_filter(anchors, 600, 800))
    # empty result and tiny inputs
    tiny = np.float32([[0, 0, 3, 3], [5, 5, 6, 6]])
    b, i = bboxes_clip_filter(g(tiny), 0, 600, 800, 16)
    assert b.shape == (0, 4) and i.shape == (0,)
    # strided where(score > thr) (prediction.py:136)
    S = syn.class_scores(777, 21, rng)
    idx, cnt = ops.where_greater(g(S)[:, 5], 0.05)
    np.testing.assert_array_equal(h(idx[:int(cnt.item())]), np.nonzero(S[:, 5] > np.float32(0.05))[0])


def test_pairwise_iou():
    from tf_eager_object_detection_amd.utils.bbox_tf import pairwise_iou
    rng = np.random.default_rng(13)
    a = syn.random_boxes(1000, (600, 800), rng)
    b = syn.random_boxes(37, (600, 800), rng)
    b[3] = a[5]
    close(h(pairwise_iou(g(a), g(b))), co.pairwise_iou(a, b), 0, 'pairwise_iou')
    anchors = co.fpn_anchors((800, 1333))
    big = h(pairwise_iou(g(anchors), g(b)))
    close(big, co.pairwise_iou(anchors, b), 0, 'pairwise_iou 267069 x 37')
    assert pairwise_iou(g(a[:0]), g(b)).shape == (0, 37)


def test_rpn_fg_softmax():
    rng = np.random.default_rng(14)
    lg = rng.normal(0, 3, (267069, 2)).astype(np.float32)
    close(h(ops.rpn_fg_softmax(g(lg), 3, ops.RPN_LAYOUT_FPN)), co.rpn_fg_fpn(lg), 1, 'rpn fg softmax')
    lg = rng.normal(0, 3, (4200, 18)).astype(np.float32)
    close(h(ops.rpn_fg_softmax(g(lg), 9, ops.RPN_LAYOUT_FRCNN)), co.rpn_fg_frcnn(lg, 9), 1, 'rpn fg softmax')
    # KA12
    one = np.float32([[1.0, 2.0, 3.0, 0.5, 2.0, 7.0]])
    np.testing.assert_allclose(h(ops.rpn_fg_softmax(g(one), 3, ops.RPN_LAYOUT_FRCNN)), on.rpn_fg_scores_frcnn(one, 3),
                               atol=1e-7)


# ----------------------------------------------------------------------------------- NMS ------
def _nms_gpu(boxes, scores, k, thr):
    idx, cnt = ops.nms(g(boxes), g(scores), k, thr)
    return h(idx[:int(cnt.item())])


def test_nms_known_answers():
    box = np.float32([[0, 0, 10, 10]])
    assert _nms_gpu(np.repeat(box, 4, 0), np.float32([.1, .9, .5, .3]), 10, 0.5).tolist() == [1]
    b = np.float32([[0, 0, 10, 10], [0, 0, 10, 5]])
    assert _nms_gpu(b, np.float32([.9, .8]), 10, 0.5).tolist() == [0, 1]       # IoU == thr: strict >
    assert _nms_gpu(b, np.float32([.9, .8]), 10, 0.49).tolist() == [0]
    z = np.float32([[0, 0, 10, 10], [5, 5, 5, 9], [5, 5, 5, 9]])
    assert _nms_gpu(z, np.float32([.9, .8, .7]), 10, 0.1).tolist() == [0, 1, 2]  # zero area
    far = np.float32([[i * 100, 0, i * 100 + 10, 10] for i in range(6)])
    assert _nms_gpu(far, np.float32([.5, .9, .9, .1, .7, .9]), 4, 0.5).tolist() == [1, 2, 5, 4]
    assert _nms_gpu(b[:, [2, 3, 0, 1]], np.float32([.9, .8]), 10, 0.49).tolist() == [0]
    assert _nms_gpu(np.zeros((0, 4), np.float32), np.zeros(0, np.float32), 5, 0.5).size == 0
    # NaN / -inf scores are never candidates (score > lowest-float is false)
    s = np.float32([0.5, np.nan, -np.inf, 0.7])
    assert _nms_gpu(far[:4], s, 4, 0.5).tolist() == [3, 0]
    # TensorFlow's published unit-test vectors (non_max_suppression_op_test.cc; third-party known answers, see
    # tests/test_oracle.py::test_tf_published_nms_vectors) through the HIP path
    tb = np.float32([[0, 0, 1, 1], [0, 0.1, 1, 1.1], [0, -0.1, 1, 0.9], [0, 10, 1, 11], [0, 10.1, 1, 11.1], [0, 100, 1, 101]])
    ts = np.float32([0.9, 0.75, 0.6, 0.95, 0.5, 0.3])
    assert _nms_gpu(tb, ts, 3, 0.5).tolist() == [3, 0, 5]
    assert _nms_gpu(np.float32([[1, 1, 0, 0], [0, 0.1, 1, 1.1], [0, 0.9, 1, -0.1], [0, 10, 1, 11], [1, 10.1, 0, 11.1],
                                [1, 101, 0, 100]]), ts, 3, 0.5).tolist() == [3, 0, 5]
    assert _nms_gpu(tb, ts, 2, 0.5).tolist() == [3, 0]
    assert _nms_gpu(tb, ts - np.float32(5.0), 6, 0.5).tolist() == [3, 0, 5]
    assert _nms_gpu(np.tile(np.float32([[0, 0, 1, 1]]), (10, 1)), np.full(10, 0.9, np.float32), 3, 0.5).tolist() == [0]


@pytest.mark.parametrize('n,k,thr,kind', [
    (1, 1, 0.7, 'distinct'), (63, 300, 0.7, 'distinct'), (64, 10, 0.3, 'tied'), (65, 65, 0.5, 'distinct'),
    (4096, 300, 0.7, 'tied'), (4097, 4097, 0.7, 'distinct'), (5000, 2000, 0.7, 'clustered'),
    (20000, 1000, 0.7, 'clustered'), (20000, 20000, 0.5, 'tied'), (37800, 300, 0.7, 'clustered'),
])
def test_nms_index_exact(n, k, thr, kind):
    rng = np.random.default_rng(n + k)
    boxes = syn.random_boxes(n, (800, 1333), rng, 16, 400)
    if kind == 'distinct':
        scores = syn.scores_distinct(n, rng)
    elif kind == 'tied':
        scores = syn.scores_tied(n, rng, 2)
    else:
        scores = syn.scores_clustered(boxes, (800, 1333), rng)
    want, stats = co.nms(boxes, scores, k, thr, True)
    got = _nms_gpu(boxes, scores, k, thr)
    np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize('n,step,k', [(40, 1, 40), (200, 1, 200), (1500, 1, 1000), (1500, 1, 333), (1400, 7, 1000), (1536, 3, 100), (3000, 1, 2000)])
def test_nms_ladders_of_any_depth(n, step, k):
    """LADDERS: box i overlaps box i +- 1 (IoU 0.74 > 0.7) and nothing else (IoU 0.54 with i +- 2), so greedy NMS keeps every other
    box of a ladder and the decision about its j-th box hangs on all j - 1 before it -- dependency chains of depth n / step through
    the score order (`step` ladders interleaved), the opposite of the sparse lists a proposal stage sees.  K cuts inside a
    64-block; n > 1536 continues in further chunks.  Kept indices exact against the oracle.  (Round 6 tried the first chunk's walk
    as a parallel fixed-point iteration -- kept[j] = present[j] and no earlier kept i overlapping j, settled after depth + 1
    rounds; 2 rounds on the bench's lists, 6-9 on clustered scores --: exact on these cases, and no faster: 18.2 us either way,
    the launch is its LDS staging, prologue and output tail, not the walk.  Not adopted.)"""
    rng = np.random.default_rng(n * 31 + step)
    w_, h_ = np.float32(100.0), np.float32(60.0)
    boxes = np.zeros((n, 4), np.float32)
    for i in range(n):
        lad, j = i % step, i // step
        x0 = np.float32(15.0 * j)                             # shift 0.15 w: IoU(i, i + 1) = 0.85 / 1.15, IoU(i, i + 2) = 0.7 / 1.3
        y0 = np.float32(200.0 * lad)                          # ladders far apart
        boxes[i] = [x0, y0, x0 + w_, y0 + h_]
    scores = np.linspace(0.99, 0.01, n).astype(np.float32)    # score order = index order
    perm = rng.permutation(n)                                 # memory order is not score order
    boxes, scores = boxes[perm], scores[perm]
    want, _ = co.nms(boxes, scores, k, 0.7, True)
    got = _nms_gpu(boxes, scores, k, 0.7)
    np.testing.assert_array_equal(got, want)
    # (greedy keeps every other box of each ladder until K are kept)
    assert len(want) == min(k, sum((n - lad + step - 1) // step // 2 + ((n - lad + step - 1) // step) % 2 for lad in range(step)))


def test_nms_dense_suppression_many_chunks():
    # nearly identical boxes: almost everything is suppressed, the scan has to walk many 4096-chunks
    rng = np.random.default_rng(5)
    n = 30000
    base = np.float32([100, 100, 300, 260])
    boxes = base + rng.uniform(-3, 3, (n, 4)).astype(np.float32)
    boxes[::1000] += np.float32([500, 300, 500, 300])           # a few far-away survivors
    scores = syn.scores_distinct(n, rng)
    want, stats = co.nms(boxes, scores, 100, 0.5, True)
    assert stats[0] > 8192                                      # really needs > 2 chunks
    np.testing.assert_array_equal(_nms_gpu(boxes, scores, 100, 0.5), want)


def test_nms_massive_ties_fall_back_to_full_order():
    # every score identical: the radix select cannot split the boundary bin -> chunk 0 is empty and
    # the full (score desc, index asc) order decides; also a two-value plateau around the threshold
    rng = np.random.default_rng(8)
    n = 10000
    boxes = syn.random_boxes(n, (800, 1333), rng, 16, 200)
    scores = np.full(n, 0.5, np.float32)
    np.testing.assert_array_equal(_nms_gpu(boxes, scores, 300, 0.5), co.nms(boxes, scores, 300, 0.5))
    scores[rng.permutation(n)[:200]] = 0.75
    np.testing.assert_array_equal(_nms_gpu(boxes, scores, 500, 0.5), co.nms(boxes, scores, 500, 0.5))
    # near-ties: 6000 scores inside one 24-bit key prefix (differences of a few ulp)
    base = np.float32(0.6)
    scores = (base + np.float32(2 ** -24) * rng.integers(0, 200, n).astype(np.float32)).astype(np.float32)
    np.testing.assert_array_equal(_nms_gpu(boxes, scores, 400, 0.5), co.nms(boxes, scores, 400, 0.5))


@pytest.mark.parametrize('shape,k,kind', [((800, 1333), 1000, 'distinct'), ((800, 1333), 1000, 'clustered'),
                                          ((800, 1333), 300, 'random_init'), ((1333, 1333), 1000, 'tied'),
                                          ((200, 320), 2000, 'distinct')])
def test_fpn_proposals_fused_stage(shape, k, kind):
    """odet_fpn_proposals = _get_anchors + softmax[:,1] + RegionProposal + _assign_levels
    (base_fpn_model.py:220-224,256) in one call: kept anchor indices, level permutation and counts
    bit-exact, boxes within 1e-4, in the sync-free mode and (if that reports "not done") the exact one."""
    from tf_eager_object_detection_amd.utils.anchor_generator import fpn_level_tables
    rng = np.random.default_rng(4321)
    anchors = co.fpn_anchors(shape)
    n = anchors.shape[0]
    deltas = syn.rpn_deltas(n, rng, 0.1)
    if kind == 'random_init':
        logits = rng.normal(0, 0.01, (n, 2)).astype(np.float32)
    else:
        prob = {'distinct': syn.scores_distinct, 'tied': lambda n_, r: syn.scores_tied(n_, r, 4)}.get(kind)
        prob = prob(n, rng) if prob else syn.scores_clustered(anchors, shape, rng)
        logits = syn.logits_from_prob(prob, rng)
    fh, fw, wh = fpn_level_tables(shape, syn.FPN_STRIDES, syn.FPN_BASE_SIZES, syn.FPN_SCALES, syn.FPN_RATIOS)
    fg = co.rpn_fg_fpn(logits)
    want_rois, want_idx = co.region_proposal(deltas, anchors, fg, shape, k, 0.7)
    lv, perm, cnt_lv = co.assign_levels(want_rois)
    done = torch.zeros(1, dtype=torch.int32, device='cuda')
    for mode_done in (done, None):
        rois, idx, cnt, (srois, slv, sperm, scnt) = ops.fpn_proposals(
            g(logits), g(deltas), fh, fw, syn.FPN_STRIDES, wh, shape, k, 0.7, M0, S1, min_level=2, max_level=5,
            blind_chunks=1, done=mode_done)
        if mode_done is not None and int(done.item()) == 0:
            continue                      # chunk 0 was not enough: the exact mode must deliver
        m = int(cnt.item())
        assert m == len(want_idx)
        np.testing.assert_array_equal(h(idx[:m]), want_idx)
        close(h(rois[:m]), want_rois, 1, 'region proposal rois')
        np.testing.assert_array_equal(h(sperm[:m]), perm)
        np.testing.assert_array_equal(h(slv[:m]) + 2, lv[perm])
        np.testing.assert_array_equal(h(scnt), cnt_lv)
        close(h(srois[:m]), want_rois[perm], 1, 'region proposal rois')


def test_nms_sync_free_mode_reports_completion():
    rng = np.random.default_rng(6)
    n = 20000
    boxes = syn.random_boxes(n, (800, 1333), rng, 16, 300)
    scores = syn.scores_distinct(n, rng)
    done = torch.zeros(1, dtype=torch.int32, device='cuda')
    idx, cnt = ops.nms(g(boxes), g(scores), 300, 0.7, blind_chunks=2, done=done)
    assert int(done.item()) == 1
    np.testing.assert_array_equal(h(idx[:int(cnt.item())]), co.nms(boxes, scores, 300, 0.7))
    # ask for more than two chunks can deliver -> done == 0, prefix still exact
    done.zero_()
    idx, cnt = ops.nms(g(boxes), g(scores), 20000, 0.7, blind_chunks=1, done=done)
    assert int(done.item()) == 0
    m = int(cnt.item())
    np.testing.assert_array_equal(h(idx[:m]), co.nms(boxes, scores, 20000, 0.7)[:m])


@pytest.mark.parametrize('shape,k,kind', [((800, 1333), 1000, 'distinct'), ((800, 1333), 1000, 'clustered'),
                                          ((800, 1333), 2000, 'clustered'), ((1333, 1333), 1000, 'tied')])
def test_region_proposal_full_size(shape, k, kind):
    """model/region_proposal.py at BASELINE sizes: 267 069 / 446 118 anchors, NMS over all of them."""
    from tf_eager_object_detection_amd.model.region_proposal import RegionProposal
    rng = np.random.default_rng(1234)
    anchors = co.fpn_anchors(shape)
    n = anchors.shape[0]
    deltas = syn.rpn_deltas(n, rng, 0.1)
    scores = {'distinct': syn.scores_distinct, 'tied': lambda n_, r: syn.scores_tied(n_, r, 4)}.get(kind, None)
    scores = scores(n, rng) if scores else syn.scores_clustered(anchors, shape, rng)
    layer = RegionProposal(num_anchors=3, num_post_nms_test=k, num_post_nms_train=2000, nms_iou_threshold=0.7,
                           target_means=M0, target_stds=S1)
    inputs = (g(deltas), g(anchors), g(scores), list(shape))
    rois_p, idx_p, cnt = layer.padded(inputs, training=False)
    want_rois, want_idx, stats = co.region_proposal(deltas, anchors, scores, shape, k, 0.7, return_stats=True)
    m = int(cnt.item())
    np.testing.assert_array_equal(h(idx_p[:m]), want_idx)          # kept anchor indices: bit-exact
    close(h(rois_p[:m]), want_rois, 1, 'RegionProposal layer')
    rois = layer(inputs, training=False)
    assert rois.shape == (len(want_idx), 4)
    close(h(rois), want_rois, 1, 'RegionProposal layer')


# ------------------------------------------------------------------------------ RoI pooling ----
def _feat(hw, c, rng):
    return rng.standard_normal((1, hw[0], hw[1], c), dtype=np.float32)


def test_roi_pooling_layers_match_oracle_bit_exact():
    from tf_eager_object_detection_amd.model.roi_pooling import (RoiPoolingCropAndResize, RoiPoolingCropAndResize2,
                                                                 RoiPoolingRoiAlign, crop_and_resize, roi_align)
    rng = np.random.default_rng(21)
    feat = _feat((38, 50), 64, rng)
    rois = syn.random_boxes(200, (600, 800), rng, 8, 700)
    rois[0] = [0, 0, 799, 599]
    rois[1] = [-50, -30, 120, 90]            # partly outside -> extrapolated zeros
    rois[2] = [300, 300, 300, 300]           # degenerate
    rois[3] = [700, 500, 1200, 900]          # beyond the map
    fg, rg = g(feat), g(rois)
    for flag in (True, False):
        got = h(RoiPoolingCropAndResize(7, flag)((fg, rg, 16)))
        want = co.roi_pool(feat, rois, stride=16, pool=7, max_pool=flag)
        np.testing.assert_array_equal(got, want)
    got = h(RoiPoolingCropAndResize2(7)((fg, rg, [600, 800])))
    np.testing.assert_array_equal(got, co.roi_pool(feat, rois, image_shape=(600, 800), pool=7))
    got = h(RoiPoolingRoiAlign(7)((fg, rg, 16)))
    close(got, co.roi_align(feat, rois, 16, 7), 0, 'RoiPoolingRoiAlign')
    # free functions (feature-map coordinates)
    fb = rois / np.float32(16)
    close(h(roi_align(fg, g(fb), 7)), on.roi_align(feat, fb, 7), 0, 'roi_align')
    close(h(crop_and_resize(fg, g(fb[:20]), torch.zeros(20, dtype=torch.int32), 14)),
          on.crop_and_resize_tp(feat, fb[:20], np.zeros(20, np.int32), 14), 0, 'crop_and_resize (tensorpack)')
    close(h(crop_and_resize(fg, g(fb[:20]), torch.zeros(20, dtype=torch.int32), 6, pad_border=False)),
          on.crop_and_resize_tp(feat, fb[:20], np.zeros(20, np.int32), 6, pad_border=False), 0,
          'crop_and_resize (tensorpack, no pad)')
    # KA8: identity crop through the P x P (no pool) path: box = whole map
    ident = h(ops.roi_pool([g(feat[:, :7, :7])], g(np.float32([[0, 0, 6, 6]])), None, ops.ROI_NORM_STRIDE, 7,
                           ops.ROI_POOL_NONE, strides=[1.0]))
    np.testing.assert_array_equal(ident[0], feat[0, :7, :7])
    assert RoiPoolingCropAndResize(7)((fg, g(rois[:0]), 16)).shape == (0, 7, 7, 64)


@pytest.mark.parametrize('C,n', [(4, 40), (256, 40), (512, 40), (1024, 40), (512, 2100), (768, 2050), (260, 40)])
def test_roi_pool_channel_counts(C, n):
    """channel counts around the kernel's 256-channel slices: one slice; several slices with one workgroup per
    (RoI, slice) (launches of fewer than 2048 RoIs) or one workgroup per RoI looping over them (n >= 2048), both with
    the register carry between neighbouring bins; a count that is not a multiple of 256 (the plain walk); float16 maps
    take the same launch shapes."""
    rng = np.random.default_rng(C + n)
    feat = _feat((20, 30), C, rng)
    rois = syn.random_boxes(n, (320, 480), rng, 8, 300)
    for pool in (ops.ROI_POOL_MAX2, ops.ROI_POOL_NONE):
        got = h(ops.roi_pool([g(feat)], g(rois), None, ops.ROI_NORM_STRIDE, 7, pool, strides=[16.0]))
        np.testing.assert_array_equal(got, co.roi_pool(feat, rois, stride=16, pool=7, max_pool=pool == ops.ROI_POOL_MAX2,
                                                       threads=8))
    if C % 8 == 0 and n <= 100:
        f16 = feat.astype(np.float16)
        got = ops.roi_pool([g(f16)], g(rois), None, ops.ROI_NORM_STRIDE, 7, ops.ROI_POOL_MAX2, strides=[16.0])
        want = co.roi_pool(f16.astype(np.float32), rois, stride=16, pool=7, max_pool=True).astype(np.float16)
        np.testing.assert_array_equal(h(got).view(np.uint16), want.view(np.uint16))


def test_fpn_roi_path_full_size():
    """assign_levels + multi-level RoI kernel at config-3 size: 1000 RoIs, P2..P5 x 256 channels."""
    from tf_eager_object_detection_amd.model.roi_pooling import roi_pooling_fpn_levels
    rng = np.random.default_rng(1234)
    shape = (800, 1333)
    shapes = syn.fpn_level_shapes(shape)[:4]
    feats = syn.features(shapes, 256, rng)
    rois = syn.random_boxes(1000, shape, rng, 16, 800)
    rois[:5] = [[0, 0, 112, 112], [0, 0, 224, 224], [0, 0, 448, 448], [10, 10, 10, 200], [0, 0, 1332, 799]]
    sorted_rois, lvl, perm, counts = ops.assign_levels(g(rois), 2, 5)
    wl, wperm, wcnt = co.assign_levels(rois)
    np.testing.assert_array_equal(h(perm), wperm)                       # bit-exact indices
    np.testing.assert_array_equal(h(counts), wcnt)
    np.testing.assert_array_equal(h(lvl) + 2, wl[wperm])
    np.testing.assert_array_equal(h(sorted_rois), rois[wperm])
    out = h(roi_pooling_fpn_levels([g(f) for f in feats], sorted_rois, lvl, shape, 7))
    rois_list, _, _ = on.assign_levels(rois)
    want = np.concatenate([co.roi_pool(f, r, image_shape=shape, pool=7, threads=8)
                           for f, r in zip(feats, rois_list) if r.shape[0]], axis=0)
    np.testing.assert_array_equal(out, want)                            # IEEE-only kernel: bit-identical


def test_fpn_roi_path_full_size_float16_maps_config5():
    """BASELINE config 5 at FULL size: 1333 x 1333, 1000 proposals, P2..P5 x 256 channels, FLOAT16 feature maps (float32
    boxes and lerps, float16 in / out): the multi-level RoI kernel against the oracle's crops of the float16 values
    widened to float32 and rounded to float16 once -- bit-identical; the same through the spatial processing order."""
    rng = np.random.default_rng(4321)
    shape = (1333, 1333)
    shapes = syn.fpn_level_shapes(shape)[:4]
    feats16 = [f.astype(np.float16) for f in syn.features(shapes, 256, rng)]
    rois = syn.random_boxes(1000, shape, rng, 16, 1200)
    rois[:4] = [[0, 0, 112, 112], [0, 0, 448, 448], [0, 0, 1332, 1332], [600, 10, 610, 900]]
    lv, perm, cnt = co.assign_levels(rois)
    srois, slv = rois[perm], (lv[perm] - 2).astype(np.int32)
    want = np.concatenate([co.roi_pool(feats16[l].astype(np.float32), srois[slv == l], image_shape=shape, pool=7, threads=8)
                           for l in range(4) if np.any(slv == l)], axis=0).astype(np.float16)
    for with_order in (False, True):
        order = ops.roi_order(g(srois), g(slv), shape) if with_order else None
        got = ops.roi_pool([g(f) for f in feats16], g(srois), g(slv), ops.ROI_NORM_IMAGE, 7, ops.ROI_POOL_MAX2,
                           image_shape=shape, order=order)
        assert got.dtype == torch.float16
        np.testing.assert_array_equal(h(got).view(np.uint16), want.view(np.uint16))


def test_assign_levels_boundaries_and_count_dev():
    def sq(s):
        return [0, 0, s, s]
    r = np.float32([sq(112), sq(224), sq(448), sq(10), sq(2000), [5, 5, 5, 50], [50, 5, 5, 5], sq(223.9), sq(111.9)])
    s, lvl, perm, counts = ops.assign_levels(g(r), 2, 5)
    assert h(perm).tolist() == [3, 5, 6, 8, 0, 7, 1, 2, 4] and h(counts).tolist() == [4, 2, 1, 2]
    cnt = torch.tensor([4], dtype=torch.int32, device='cuda')
    s, lvl, perm, counts = ops.assign_levels(g(r), 2, 5, count_dev=cnt)
    assert h(counts).tolist() == [1, 1, 1, 1] and h(perm)[:4].tolist() == [3, 0, 1, 2]
    rng = np.random.default_rng(3)
    big = syn.random_boxes(8192, (800, 1333), rng, 4, 1200)
    _, _, perm, counts = ops.assign_levels(g(big), 2, 5)
    wl, wperm, wcnt = co.assign_levels(big)
    np.testing.assert_array_equal(h(perm), wperm)


# ------------------------------------------------------------------------------- post-ops -----
def _check_post(got, want):
    gb, gl, gs = got
    wb, wl, ws = want
    if wb is None:
        assert gb is None and gl is None and gs is None
        return
    assert gl.dtype == torch.int32
    np.testing.assert_array_equal(h(gl), wl)                     # labels exact (same order: score desc)
    close(h(gs), ws, 0, 'post-ops scores')
    close(h(gb), wb, 1, 'post-ops boxes')


@pytest.mark.parametrize('R,ncls,mpc,mpi,sthr', [(300, 21, 50, 50, 0.0), (1000, 21, 50, 50, 0.0),
                                                 (1000, 81, 100, 300, 0.0), (1000, 21, 50, 150, 0.05),
                                                 (7, 21, 50, 50, 0.0), (1000, 81, 100, 100, 0.05)])
def test_post_ops_prediction(R, ncls, mpc, mpi, sthr):
    from tf_eager_object_detection_amd.model.prediction import post_ops_prediction
    rng = np.random.default_rng(R + ncls)
    shape = (800, 1333)
    S = syn.class_scores(R, ncls, rng)
    D = syn.class_deltas(R, ncls, rng)
    rois = syn.random_boxes(R, shape, rng, 16, 600)
    got = post_ops_prediction(g(S), g(D), g(rois), list(shape), M0, S2, mpc, mpi, 0.3, sthr, 16, num_classes=ncls)
    want = co.post_ops(S, D, rois, shape, M0, S2, mpc, mpi, 0.3, sthr, 16, ncls)
    _check_post(got, want)


def test_post_ops_large_r_dense_suppression_and_record():
    """More than 1024 RoIs (LDS bitonic path), heavily overlapping RoIs with zero deltas (the per-class
    NMS needs many 128-candidate rounds and the cross test against earlier rounds), and the fused
    detection record of odet_post_ops_record == odet_pack_detections of the padded outputs."""
    from tf_eager_object_detection_amd.model.prediction import post_ops_prediction
    from tf_eager_object_detection_amd import parallel
    rng = np.random.default_rng(21)
    shape = (800, 1333)
    R, ncls = 2500, 21
    S = syn.class_scores(R, ncls, rng)
    D = syn.class_deltas(R, ncls, rng)
    rois = syn.random_boxes(R, shape, rng, 16, 600)
    got = post_ops_prediction(g(S), g(D), g(rois), list(shape), M0, S2, 100, 300, 0.3, 0.0, 16, num_classes=ncls)
    _check_post(got, co.post_ops(S, D, rois, shape, M0, S2, 100, 300, 0.3, 0.0, 16, ncls))
    # dense: 1000 RoIs in 12 tight clusters, deltas ~0 -> almost everything suppressed inside a cluster
    R = 1000
    centers = rng.uniform(150, 600, (12, 2)).astype(np.float32)
    cid = rng.integers(0, 12, R)
    ctr = centers[cid] + rng.uniform(-6, 6, (R, 2)).astype(np.float32)
    half = (60 + rng.uniform(-4, 4, (R, 2))).astype(np.float32)
    rois = np.concatenate([ctr - half, ctr + half], axis=1).astype(np.float32)
    S = syn.class_scores(R, ncls, rng)
    D = (syn.class_deltas(R, ncls, rng) * np.float32(0.02)).astype(np.float32)
    for mpc, mpi in ((50, 50), (300, 1000)):
        got = post_ops_prediction(g(S), g(D), g(rois), list(shape), M0, S2, mpc, mpi, 0.3, 0.0, 16, num_classes=ncls)
        _check_post(got, co.post_ops(S, D, rois, shape, M0, S2, mpc, mpi, 0.3, 0.0, 16, ncls))
    # record variant
    rec = torch.empty(50 * 6 + 1, dtype=torch.float32, device='cuda')
    ob, ol, os_, cnt = ops.post_ops(g(S), g(D), g(rois), shape, M0, S2, 50, 50, 0.3, 0.0, 16, ncls, record=rec)
    want = parallel.pack_detections(ob, ol, os_, cnt, 50)
    np.testing.assert_array_equal(h(rec), h(want))
    assert float(rec[-1]) == float(cnt.item())


def test_post_ops_edge_cases():
    from tf_eager_object_detection_amd.model.prediction import post_ops_prediction
    rng = np.random.default_rng(8)
    shape = (600, 800)
    R = 64
    S = syn.class_scores(R, 81, rng)
    D = syn.class_deltas(R, 81, rng)
    rois = syn.random_boxes(R, shape, rng, 16, 400)
    # nothing survives -> (None, None, None) (prediction.py:153-154)
    assert post_ops_prediction(g(S), g(D), g(rois), list(shape), M0, S2, 5, 5, 0.3, 1.5, 16) == (None, None, None)
    # default num_classes=21 on 81 columns only visits classes 1..20 (reference quirk)
    got = post_ops_prediction(g(S), g(D), g(rois), list(shape), M0, S2, 50, 150, 0.3, 0.0, 16)
    _check_post(got, co.post_ops(S, D, rois, shape, M0, S2, 50, 150, 0.3, 0.0, 16, 21))
    assert int(got[1].max()) <= 20
    # target_means / target_stds None -> defaults (prediction.py:128-131)
    got = post_ops_prediction(g(S), g(D), g(rois), list(shape), None, None, 50, 150, 0.3, 0.0, 16)
    _check_post(got, co.post_ops(S, D, rois, shape, None, None, 50, 150, 0.3, 0.0, 16, 21))
    # min-edge filter removes everything small
    small = np.float32([[10, 10, 14, 14]] * R)
    Dz = np.zeros_like(D)
    assert post_ops_prediction(g(S), g(Dz), g(small), list(shape), M0, S2, 5, 5, 0.3, 0.0, 16) == (None, None, None)
    # tied scores: every RoI identical score per class -> lower RoI index wins
    St = np.full((R, 21), 1.0 / 21, np.float32)
    got = post_ops_prediction(g(St), g(D[:, :21]), g(rois), list(shape), M0, S2, 50, 150, 0.3, 0.0, 16)
    _check_post(got, co.post_ops(St, D[:, :21], rois, shape, M0, S2, 50, 150, 0.3, 0.0, 16, 21))


@pytest.mark.parametrize('R,mpc,mpi,sthr,decimals', [(1000, 50, 50, 0.05, 2), (1000, 50, 50, 0.0, 1), (1000, 5, 5, 0.05, 2),
                                                      (1000, 50, 64, 0.05, 2), (200, 50, 50, 0.2, 2), (1000, 51, 64, 0.0, 2),
                                                      (40, 50, 50, 0.0, 3), (1000, 50, 65, 0.05, 2), (1500, 50, 50, 0.05, 2),
                                                      (2500, 30, 100, 0.0, 1), (4096, 50, 50, 0.1, 2)])
def test_post_ops_ties_through_the_merge(R, mpc, mpi, sthr, decimals):
    """Scores quantised to a few distinct values: the per-class order and the image's top-k are decided by the tie rule
    (row / position order), and the max_per_image-th best score is shared by many entries.  One launch (k_postops): a class
    workgroup sorts the rows that pass its filters (1024 keys, or the next power of two of R for R > 1024: the last three
    cases), the last workgroup merges by a radix select of the max_per_image-th key with ties in position order."""
    from tf_eager_object_detection_amd.model.prediction import post_ops_prediction
    rng = np.random.default_rng(R + mpc + mpi)
    shape = (800, 1333)
    S = np.round(syn.class_scores(R, 21, rng), decimals).astype(np.float32)
    D = (syn.class_deltas(R, 21, rng) * np.float32(0.3)).astype(np.float32)
    rois = syn.random_boxes(R, shape, rng, 16, 300)
    got = post_ops_prediction(g(S), g(D), g(rois), list(shape), M0, S2, mpc, mpi, 0.3, sthr, 16, num_classes=21)
    _check_post(got, co.post_ops(S, D, rois, shape, M0, S2, mpc, mpi, 0.3, sthr, 16, 21))


def test_predict_after_roi_matches_oracle():
    from tf_eager_object_detection_amd.model.prediction import predict_after_roi
    rng = np.random.default_rng(9)
    R = 200
    S = syn.class_scores(R, 21, rng, 3.0)
    D = syn.class_deltas(R, 21, rng)
    rois = syn.random_boxes(R, (600, 800), rng, 16, 400)
    gb, gl, gs = predict_after_roi(g(S), g(D), g(rois), [600, 800], M0, S2, 5, 20, 0.3, 0.3)
    wb, wl, ws = on.predict_after_roi(S, D, rois, (600, 800), M0, S2, 5, 20, 0.3, 0.3)
    np.testing.assert_array_equal(h(gl), wl)
    close(h(gs), ws, 0, 'post-ops scores')
    close(h(gb), wb, 1, 'post-ops boxes')


# ---------------------------------------------------------------------- size-independent ------
def test_properties_at_full_size():
    """Checks that do not need the oracle: NMS output is sorted by score, pairwise non-overlapping
    above thr, maximal (every rejected top candidate overlaps a kept one), idempotent."""
    rng = np.random.default_rng(77)
    n = 267069
    boxes = syn.random_boxes(n, (800, 1333), rng, 8, 600)
    scores = syn.scores_clustered(boxes, (800, 1333), rng)
    idx, cnt = ops.nms(g(boxes), g(scores), 1000, 0.7)
    k = h(idx[:int(cnt.item())]).astype(np.int64)
    assert len(k) == 1000 and len(np.unique(k)) == 1000
    assert np.all(np.diff(scores[k]) <= 0)
    kb = boxes[k]
    iou = co.pairwise_iou(kb - np.float32([0, 0, 1, 1]), kb - np.float32([0, 0, 1, 1]))   # (+1 cancels)
    np.fill_diagonal(iou, 0)
    assert iou.max() <= 0.7 + 1e-6
    # idempotence: NMS of the kept set keeps everything, same order
    idx2, cnt2 = ops.nms(g(kb), g(scores[k]), 1000, 0.7)
    assert h(idx2[:int(cnt2.item())]).tolist() == list(range(1000))


# ------------------------------------------------------------------ throughput arrangement ----
def test_stream_pool_matches_single_path_and_oracle():
    """FpnStreamPool (native executor, several images in flight on their own streams) must produce,
    for every slot, exactly what the single-stream FpnHotPath produces -- and that is the oracle's."""
    from tf_eager_object_detection_amd.pipeline import FpnHotPath, FpnStreamPool, synthetic_fpn_inputs
    shape, K, ncls, ch = (320, 480), 300, 21, 64
    sets = [synthetic_fpn_inputs(shape, ncls, K, channels=ch, seed=100 + i) for i in range(3)]
    ref = FpnHotPath(shape, ncls, K, channels=ch)
    want = []
    for host, dev in sets:
        ref.step(dev['rpn_logits'], dev['rpn_deltas'], dev['feats'], dev['cls_scores'], dev['cls_deltas'])
        torch.cuda.synchronize()
        assert int(ref.nms_done.item()) == 1
        want.append((ref.record.clone(), ref.roi_features.clone(), ref.roi_idx.clone(), int(ref.roi_count.item())))
    # oracle check of image 0's proposals (the rest is covered stage by stage elsewhere)
    host = sets[0][0]
    fg = co.rpn_fg_fpn(host['rpn_logits'])
    _, idx = co.region_proposal(host['rpn_deltas'], co.fpn_anchors(shape), fg, shape, K, 0.7)
    np.testing.assert_array_equal(h(want[0][2][:want[0][3]]), idx)
    pool = FpnStreamPool(3, shape, ncls, K, ch)
    try:
        for k, (host, dev) in enumerate(sets):
            pool.bind(k, dev['rpn_logits'], dev['rpn_deltas'], dev['feats'], dev['cls_scores'], dev['cls_deltas'])
        for rnd in range(5):                      # several rounds: slots are re-used back to back
            for k in range(3):
                pool.submit(k)
        pool.wait()
        torch.cuda.synchronize()
        for k in range(3):
            slot = pool.slots[k]
            assert int(slot.nms_done.item()) == 1
            assert torch.equal(slot.record, want[k][0])
            assert torch.equal(slot.roi_features, want[k][1])
            assert int(slot.roi_count.item()) == want[k][3]
            assert torch.equal(slot.roi_idx[:want[k][3]], want[k][2][:want[k][3]])
        with pytest.raises(ValueError):
            pool.bind(0, sets[0][1]['rpn_logits'][:-1], sets[0][1]['rpn_deltas'], sets[0][1]['feats'],
                      sets[0][1]['cls_scores'], sets[0][1]['cls_deltas'])
    finally:
        pool.close()


def test_stream_pool_float16_maps_match_single_path():
    """BASELINE config 5 ("fp16 feature maps") through the native executor: float16 slots, single launches and
    batched launches, give the float16 single-stream result bit for bit."""
    from tf_eager_object_detection_amd.pipeline import FpnHotPath, FpnStreamPool, synthetic_fpn_inputs
    shape, K, ncls, ch = (320, 480), 300, 21, 64
    sets = [synthetic_fpn_inputs(shape, ncls, K, channels=ch, seed=300 + i) for i in range(4)]
    for _, dev in sets:
        dev['feats'] = [f.half() for f in dev['feats']]
    ref = FpnHotPath(shape, ncls, K, channels=ch, feature_dtype=torch.float16)
    want = []
    for _, dev in sets:
        ref.step(dev['rpn_logits'], dev['rpn_deltas'], dev['feats'], dev['cls_scores'], dev['cls_deltas'])
        torch.cuda.synchronize()
        want.append((ref.record.clone(), ref.roi_features.clone()))
    assert want[0][1].dtype == torch.float16 and float(want[0][1].float().abs().max()) > 0
    for batch in (1, 4):
        pool = FpnStreamPool(4 // batch, shape, ncls, K, ch, batch=batch, feature_dtype=torch.float16)
        try:
            for k, (_, dev) in enumerate(sets):
                pool.bind(k, dev['rpn_logits'], dev['rpn_deltas'], dev['feats'], dev['cls_scores'], dev['cls_deltas'])
            if batch == 1:
                for k in range(4):
                    pool.submit(k)
            else:
                pool.submit_group(0)
            pool.wait()
            torch.cuda.synchronize()
            for k in range(4):
                assert torch.equal(pool.slots[k].roi_features, want[k][1])
                assert torch.equal(pool.slots[k].record, want[k][0])
            with pytest.raises(ValueError):          # float32 maps into a float16 slot
                pool.bind(0, sets[0][1]['rpn_logits'], sets[0][1]['rpn_deltas'], [f.float() for f in sets[0][1]['feats']],
                          sets[0][1]['cls_scores'], sets[0][1]['cls_deltas'])
        finally:
            pool.close()


def test_batched_launches_match_single_path():
    """odet_fpn_step_enqueue_batch: several images in the SAME kernel launches (blockIdx.y = image) must
    give every image exactly the single-image result, stage by stage and for the whole step."""
    from tf_eager_object_detection_amd.pipeline import FpnHotPath, FpnStreamPool, synthetic_fpn_inputs
    shape, K, ncls, ch = (320, 480), 300, 21, 64
    sets = [synthetic_fpn_inputs(shape, ncls, K, channels=ch, seed=200 + i) for i in range(6)]
    ref = FpnHotPath(shape, ncls, K, channels=ch)
    want = []
    for host, dev in sets:
        ref.step(dev['rpn_logits'], dev['rpn_deltas'], dev['feats'], dev['cls_scores'], dev['cls_deltas'])
        torch.cuda.synchronize()
        assert int(ref.nms_done.item()) == 1
        want.append([t.clone() for t in (ref.record, ref.roi_features, ref.sorted_rois, ref.roi_perm, ref.roi_level,
                                         ref.level_counts, ref.roi_count, ref.det_boxes, ref.det_labels)])
    pool = FpnStreamPool(2, shape, ncls, K, ch, batch=3)
    try:
        for k, (host, dev) in enumerate(sets):
            pool.bind(k, dev['rpn_logits'], dev['rpn_deltas'], dev['feats'], dev['cls_scores'], dev['cls_deltas'])
        for stages in ((1, 2, 4), (7,), (3, 4)):                  # stage by stage, whole step, mixed
            for h_ in pool.slots:
                h_.record.zero_(); h_.roi_features.zero_()
            for st_ in stages:
                pool.submit_group(0, stages=st_)
                pool.submit_group(1, stages=st_)
            pool.wait()
            torch.cuda.synchronize()
            for k in range(6):
                slot = pool.slots[k]
                got = (slot.record, slot.roi_features, slot.sorted_rois, slot.roi_perm, slot.roi_level,
                       slot.level_counts, slot.roi_count, slot.det_boxes, slot.det_labels)
                assert int(slot.nms_done.item()) == 1
                cnt = int(want[k][6].item())
                for name, a, b in zip(('record', 'features', 'rois', 'perm', 'level', 'level_counts', 'count',
                                       'boxes', 'labels'), got, want[k]):
                    if name in ('rois', 'perm', 'level'):
                        a, b = a[:cnt], b[:cnt]
                    assert torch.equal(a, b), (k, name, stages)
        # a single image through the same pool (launch sequence of its own) still works
        pool.submit(4)
        pool.wait(); torch.cuda.synchronize()
        assert torch.equal(pool.slots[4].record, want[4][0])
    finally:
        pool.close()


def test_roi_pool_float16_feature_maps_bit_exact():
    """BASELINE config 5: float16 feature maps into the RoI kernel (float32 boxes, float32 lerp, float16
    store).  Oracle: the same crops on the float16 values widened to float32, rounded to float16."""
    rng = np.random.default_rng(55)
    shape = (336, 336)
    shapes = syn.fpn_level_shapes(shape)[:4]
    feats16 = [rng.standard_normal((1, h_, w_, 64)).astype(np.float16) for h_, w_ in shapes]
    rois = syn.random_boxes(300, shape, rng, 8, 300)
    lv, perm, cnt = co.assign_levels(rois)
    srois, slv = rois[perm], (lv[perm] - 2).astype(np.int32)
    for mode in (ops.ROI_POOL_MAX2, ops.ROI_POOL_AVG2):
        for with_order in (False, True):
            order = ops.roi_order(g(srois), g(slv), shape) if with_order else None
            got = ops.roi_pool([g(f) for f in feats16], g(srois), g(slv), ops.ROI_NORM_IMAGE, 7, mode,
                               image_shape=shape, order=order)
            assert got.dtype == torch.float16
            want = np.concatenate([
                (co.roi_pool(feats16[l].astype(np.float32), srois[slv == l], image_shape=shape, pool=7)
                 if mode == ops.ROI_POOL_MAX2 else
                 on.tf_avg_pool_2x2(on.tf_crop_and_resize(
                     feats16[l].astype(np.float32),
                     np.stack([srois[slv == l][:, 1] / np.float32(shape[0]), srois[slv == l][:, 0] / np.float32(shape[1]),
                               srois[slv == l][:, 3] / np.float32(shape[0]), srois[slv == l][:, 2] / np.float32(shape[1])],
                              axis=1).astype(np.float32), np.zeros(int(np.sum(slv == l)), np.int32), (14, 14))))
                for l in range(4) if np.any(slv == l)], axis=0).astype(np.float16)
            np.testing.assert_array_equal(h(got).view(np.uint16), want.view(np.uint16))
    # the un-pooled 7 x 7 crop (ResNet-C4 form, roi_pooling.py:85-90) on float16 maps
    got = ops.roi_pool([g(feats16[0])], g(srois[:64]), None, ops.ROI_NORM_STRIDE, 7, ops.ROI_POOL_NONE, strides=[4.0])
    want = co.roi_pool(feats16[0].astype(np.float32), srois[:64], stride=4, pool=7, max_pool=False).astype(np.float16)
    np.testing.assert_array_equal(h(got).view(np.uint16), want.view(np.uint16))


def test_executor_reports_job_errors_and_recovers():
    """A job that fails inside a worker thread (workspace too small) must surface at wait() with the C
    ABI's error text, and the executor must keep working afterwards."""
    from tf_eager_object_detection_amd import _lib
    from tf_eager_object_detection_amd.pipeline import FpnStreamPool, synthetic_fpn_inputs
    shape, K, ncls, ch = (200, 320), 200, 21, 32
    host, dev = synthetic_fpn_inputs(shape, ncls, K, channels=ch, seed=3)
    pool = FpnStreamPool(2, shape, ncls, K, ch)
    try:
        for k in range(2):
            pool.bind(k, dev['rpn_logits'], dev['rpn_deltas'], dev['feats'], dev['cls_scores'], dev['cls_deltas'])
        good = pool.steps[1].ws_rpn_bytes
        pool.steps[1].ws_rpn_bytes = 1024                      # sabotage slot 1
        pool.submit(0)
        pool.submit(1)
        with pytest.raises(_lib.OdetError, match='workspace too small'):
            pool.wait()
        pool.steps[1].ws_rpn_bytes = good
        pool.submit(0)
        pool.submit(1)
        pool.wait()                                            # no stale error
        torch.cuda.synchronize()
        assert int(pool.slots[1].nms_done.item()) == 1 and int(pool.slots[1].det_count.item()) > 0
        # batches must agree in configuration
        pool2 = FpnStreamPool(1, shape, ncls, K, ch, batch=2)
        try:
            for k in range(2):
                pool2.bind(k, dev['rpn_logits'], dev['rpn_deltas'], dev['feats'], dev['cls_scores'], dev['cls_deltas'])
            pool2.steps[1].max_per_class = 7
            pool2.submit_group(0)
            with pytest.raises(_lib.OdetError, match='differs'):
                pool2.wait()
        finally:
            pool2.close()
    finally:
        pool.close()


def test_roi_pool_timed_events():
    rng = np.random.default_rng(3)
    feat = _feat((40, 60), 64, rng)
    rois = syn.random_boxes(50, (640, 960), rng, 16, 300)
    e0, e1 = ops.ProfEvent(), ops.ProfEvent()
    a = ops.roi_pool([g(feat)], g(rois), None, ops.ROI_NORM_STRIDE, 7, ops.ROI_POOL_MAX2, strides=[16.0])
    b = ops.roi_pool([g(feat)], g(rois), None, ops.ROI_NORM_STRIDE, 7, ops.ROI_POOL_MAX2, strides=[16.0],
                     events=(e0, e1))
    ms = e0.elapsed_ms(e1)
    assert torch.equal(a, b)
    assert 0.0 < ms < 50.0


@pytest.mark.parametrize('name,shape,channels,flag,scales', [
    ('vgg16 (BASELINE config 1 shapes)', (600, 800), 512, True, (8, 16, 32)),
    ('resnet C4 (BASELINE config 2 shapes)', (800, 1333), 1024, False, (8, 16, 32)),
    ('coco anchors', (320, 480), 64, True, (4, 8, 16, 32)),
])
def test_frcnn_hot_path_full_size(name, shape, channels, flag, scales):
    """BaseFasterRcnn.call inference branch (base_faster_rcnn_model.py:126-198) at the reference's
    shapes: 17 100 / 37 800 anchors, K = 300, VGG16 14x14+max on 512 channels / ResNet C4 7x7 on 1024."""
    from tf_eager_object_detection_amd.pipeline import FrcnnHotPath
    rng = np.random.default_rng(len(name))
    K, ncls = 300, 21
    # (the small dense case needs more than the first NMS chunk: enqueue the guarded fallback as well)
    hot = FrcnnHotPath(shape, ncls, K, channels, max_pooling_flag=flag, scales=scales, blind_chunks=4)
    A, fh, fw, n = hot.A, hot.fh, hot.fw, hot.N
    logits = rng.normal(0, 1.5, (fh * fw, 2 * A)).astype(np.float32)
    deltas = syn.rpn_deltas(n, rng, 0.1)
    feat = rng.standard_normal((1, fh, fw, channels), dtype=np.float32)
    S = syn.class_scores(K, ncls, rng)
    D = syn.class_deltas(K, ncls, rng)
    feats, boxes, labels, scores, count = hot.step(g(logits), g(deltas), g(feat), g(S), g(D))
    torch.cuda.synchronize()
    assert int(hot.nms_done.item()) == 1
    # oracle, stage by stage
    base = on.generate_anchor_base(16, (0.5, 1, 2), scales).astype(np.float32)
    anchors = on.generate_by_anchor_base_tf(base, 16, fh, fw)
    fg = on.rpn_fg_scores_frcnn(logits, A)
    want_rois, want_idx = co.region_proposal(deltas, anchors, fg, shape, K, 0.7)
    k = int(hot.roi_count.item())
    assert k == len(want_idx)
    np.testing.assert_array_equal(h(hot.roi_idx[:k]), want_idx)
    close(h(hot.rois[:k]), want_rois, 1, 'frcnn hot path rois')
    want_f = co.roi_pool(feat[0], want_rois, stride=16, pool=7, max_pool=flag)
    got_f = h(feats[:k])
    assert got_f.shape == want_f.shape
    assert np.max(np.abs(got_f - want_f)) <= 1e-4
    wb, wl, ws = co.post_ops(S[:k], D[:k], want_rois, shape, M0, S2, 50, 50, 0.3, 0.0, 16, ncls)
    m = int(count.item())
    assert m == len(ws)
    np.testing.assert_array_equal(h(labels[:m]), wl)
    close(h(boxes[:m]), wb, 1, 'frcnn hot path boxes')


def test_wide_first_nms_chunk_completes_clustered_scores_in_batched_launches():
    """odet_fpn_step_t.nms_first_chunk: trained-like clustered RPN scores need more than ~1.5 K candidates to keep
    K proposals; with a 4096-candidate first chunk, or with a second sync-free chunk per image (blind_chunks = 2),
    the batched launches complete and give the oracle's proposals; with one narrow chunk they report nms_done = 0."""
    from tf_eager_object_detection_amd.pipeline import FpnStreamPool, synthetic_fpn_inputs
    shape, K, ncls, ch = (800, 1333), 1000, 21, 8
    sets = [synthetic_fpn_inputs(shape, ncls, K, channels=ch, seed=500 + i, score_kind='clustered') for i in range(2)]
    anchors = co.fpn_anchors(shape)
    for first, blind, expect_done in ((0, 1, False), (4096, 1, True), (0, 2, True)):
        # (0, 2): the narrow shared chunk, then one sync-free fallback chunk per image inside the same call
        pool = FpnStreamPool(1, shape, ncls, K, ch, batch=2, nms_first_chunk=first, blind_chunks=blind)
        try:
            for k, (_, dev) in enumerate(sets):
                pool.bind(k, dev['rpn_logits'], dev['rpn_deltas'], dev['feats'], dev['cls_scores'], dev['cls_deltas'])
            pool.submit_group(0)
            pool.wait()
            torch.cuda.synchronize()
            for k, (host, _) in enumerate(sets):
                slot = pool.slots[k]
                done = int(slot.nms_done.item()) == 1
                assert done == expect_done
                if done:
                    fg = co.rpn_fg_fpn(host['rpn_logits'])
                    _, idx = co.region_proposal(host['rpn_deltas'], anchors, fg, shape, K, 0.7)
                    m = int(slot.roi_count.item())
                    assert m == len(idx)
                    np.testing.assert_array_equal(h(slot.roi_idx[:m]), idx)
                else:
                    # an incomplete image is reported EMPTY (odet.h, odet_nms): no stage ever runs on a partial or
                    # stale RoI list -- zero proposals, zero RoI features, zero detections, nms_done = 0
                    assert int(slot.roi_count.item()) == 0 and int(slot.det_count.item()) == 0
                    assert int(h(slot.level_counts).sum()) == 0
                    assert float(slot.roi_features.abs().max().item()) == 0.0
                    assert float(slot.record[-1].item()) == 0.0
        finally:
            pool.close()
    with pytest.raises(Exception):
        bad = FpnStreamPool(1, shape, ncls, K, ch, batch=2, nms_first_chunk=5000)
        try:
            for k, (_, dev) in enumerate(sets):
                bad.bind(k, dev['rpn_logits'], dev['rpn_deltas'], dev['feats'], dev['cls_scores'], dev['cls_deltas'])
            bad.submit_group(0)
            bad.wait()
        finally:
            bad.close()


def test_batched_full_order_chunks_match_oracle():
    """Chunks 2.. of a sync-free BATCH (the full radix sort of every image's keys and further 4096-candidate chunks in
    launches shared by the images, odet_sort_keys_desc_batch): proposals = the anchors grown e-fold with scores that fall
    along the grid (the best candidates are neighbours and suppress each other ~16 to 1), so chunk 0 + the selection's chunk 1 (~5.6 K
    candidates) cannot keep K boxes (the oracle's pop count says how many chunks it takes); with 2 sync-free chunks the
    heavy images report nms_done = 0 and are empty, with enough chunks every image gives the oracle's proposals.  One
    image has far fewer competitive scores: it finishes EARLY and skips the shared launches."""
    from tf_eager_object_detection_amd.pipeline import FpnStreamPool, synthetic_fpn_inputs
    shape, K, ncls, ch, B = (608, 800), 1000, 21, 8, 3
    sets = []
    for i in range(B):
        host, dev = synthetic_fpn_inputs(shape, ncls, K, channels=ch, seed=900 + i)
        host = dict(host); dev = dict(dev)
        d = np.zeros_like(host['rpn_deltas'])
        if i != 1:
            # boxes e times the anchors, scores falling with the anchor index (+ a little noise): the best candidates
            # are neighbours on the P2 grid and suppress each other ~16 to 1
            d[:, 2:] = 1.0
            lg = np.zeros_like(host['rpn_logits'])
            lg[:, 1] = -2e-4 * np.arange(lg.shape[0], dtype=np.float32) + np.random.default_rng(i).normal(0, 1e-3, lg.shape[0]).astype(np.float32)
            host['rpn_logits'] = lg
            dev['rpn_logits'] = g(lg)
        host['rpn_deltas'] = d
        dev['rpn_deltas'] = g(d)
        sets.append((host, dev))
    anchors = co.fpn_anchors(shape)
    wants, popped = [], []
    for host, _ in sets:
        fg = co.rpn_fg_fpn(host['rpn_logits'])
        _, idx, stats = co.region_proposal(host['rpn_deltas'], anchors, fg, shape, K, 0.7, return_stats=True)
        wants.append(idx)
        popped.append(int(stats[0]))
    assert popped[0] > 1536 + 4096 and popped[2] > 1536 + 4096 and popped[1] < 1536 + 4096, popped
    enough = 2 + (max(popped) - 1536 - 4096 + 4095) // 4096 + 1
    for blind in (2, enough):
        pool = FpnStreamPool(1, shape, ncls, K, ch, batch=B, blind_chunks=blind)
        try:
            for k, (_, dev) in enumerate(sets):
                pool.bind(k, dev['rpn_logits'], dev['rpn_deltas'], dev['feats'], dev['cls_scores'], dev['cls_deltas'])
            pool.submit_group(0)
            pool.wait()
            torch.cuda.synchronize()
            dones = [int(pool.slots[k].nms_done.item()) == 1 for k in range(B)]
            assert dones == ([False, True, False] if blind == 2 else [True, True, True]), (blind, dones, popped)
            for k in range(B):
                slot = pool.slots[k]
                m = int(slot.roi_count.item())
                if dones[k]:
                    assert m == len(wants[k])
                    np.testing.assert_array_equal(h(slot.roi_idx[:m]), wants[k])
                else:
                    assert m == 0
            if blind == 2:
                # the flagged images -- and only those -- go through the exact mode again and end with the
                # oracle's proposals and non-empty detections; nothing raises
                before = h(pool.slots[1].roi_idx).copy()
                assert pool.recover_incomplete() == [0, 2] and pool.nms_reruns == 2
                torch.cuda.synchronize()
                assert pool.nms_done_all.tolist() == [1, 1, 1] and pool.recover_incomplete() == []
                np.testing.assert_array_equal(h(pool.slots[1].roi_idx), before)
                for k in range(B):
                    slot = pool.slots[k]
                    m = int(slot.roi_count.item())
                    assert m == len(wants[k]) and int(slot.det_count.item()) > 0
                    np.testing.assert_array_equal(h(slot.roi_idx[:m]), wants[k])
                    assert float(slot.record[-1]) == float(int(slot.det_count.item()))
        finally:
            pool.close()


def test_nms_sync_free_chunks_from_selection_then_full_order():
    """Sync-free jobs: chunk 1 comes from the ranked radix selection (no sort), chunks 2.. from the full order per
    image; exhausted selections (massive ties: the boundary bin cannot be split) and dense suppression included."""
    rng = np.random.default_rng(15)
    done = torch.zeros(1, dtype=torch.int32, device='cuda')
    # (a) dense cluster: needs > 8192 candidates -> blind 2 cannot finish (prefix exact), blind 8 does
    n = 30000
    base = np.float32([100, 100, 300, 260])
    boxes = base + rng.uniform(-3, 3, (n, 4)).astype(np.float32)
    boxes[::1000] += np.float32([500, 300, 500, 300])
    scores = syn.scores_distinct(n, rng)
    want, stats = co.nms(boxes, scores, 100, 0.5, True)
    assert stats[0] > 8192
    idx, cnt = ops.nms(g(boxes), g(scores), 100, 0.5, blind_chunks=2, done=done)
    m = int(cnt.item())
    assert int(done.item()) == 0 and 0 < m <= len(want)         # (not done: the order has not been walked to its end)
    np.testing.assert_array_equal(h(idx[:m]), want[:m])
    done.zero_()
    idx, cnt = ops.nms(g(boxes), g(scores), 100, 0.5, blind_chunks=9, done=done)
    assert int(done.item()) == 1
    np.testing.assert_array_equal(h(idx[:int(cnt.item())]), want)
    # (b) every score identical: the selection is empty, chunk 1 finds nothing, the full order decides from chunk 2
    n = 10000
    boxes = syn.random_boxes(n, (800, 1333), rng, 16, 200)
    scores = np.full(n, 0.5, np.float32)
    done.zero_()
    idx, cnt = ops.nms(g(boxes), g(scores), 300, 0.5, blind_chunks=4, done=done)
    assert int(done.item()) == 1
    np.testing.assert_array_equal(h(idx[:int(cnt.item())]), co.nms(boxes, scores, 300, 0.5))
    # (c) moderate suppression finishing inside chunk 1 (selection order), K = 1000 of 60000
    n = 60000
    boxes = syn.random_boxes(n, (800, 1333), rng, 40, 400)
    scores = syn.scores_distinct(n, rng)
    want, stats = co.nms(boxes, scores, 1000, 0.3, True)
    done.zero_()
    idx, cnt = ops.nms(g(boxes), g(scores), 1000, 0.3, blind_chunks=2, done=done)
    if stats[0] <= 1536 + 4096:
        assert int(done.item()) == 1
        np.testing.assert_array_equal(h(idx[:int(cnt.item())]), want)
    else:
        m = int(cnt.item())
        np.testing.assert_array_equal(h(idx[:m]), want[:m])


@pytest.mark.parametrize('name', ['config1_vgg16_600x800', 'config2_resnet_c4_800x1333', 'config3_resnet101_fpn_800x1333',
                                  'config5_resnet101_fpn_1333x1333_81_classes'])
def test_full_size_hashes_from_the_hip_path(name):
    """SURVEY.md 8(c) full-size fixtures: the HIP hot path (through the C ABI) reproduces the committed SHA-256
    digests of every discrete output at the BASELINE.json shapes -- kept anchor indices of the NMS over all
    anchors, level assignment, detection labels (tests/golden/full_size_hashes.json, written by the oracle)."""
    import full_size_cases as fs
    from tf_eager_object_detection_amd.pipeline import FpnHotPath, FrcnnHotPath
    c = fs.CASES[name]
    inp = fs.make_inputs(name)
    want = fs.load_golden()[name]
    assert inp['num_anchors'] == want['num_anchors']
    gen = torch.Generator(device='cuda'); gen.manual_seed(5)
    if c['kind'] == 'fpn':
        hot = FpnHotPath(c['shape'], c['ncls'], c['K'], c['channels'], max_per_class=c['per_class'],
                         max_per_image=c['per_image'], blind_chunks=4)
        feats = [torch.randn((1, fh, fw, c['channels']), device='cuda', generator=gen)
                 for fh, fw in syn.fpn_level_shapes(c['shape'])[:4]]
    else:
        hot = FrcnnHotPath(c['shape'], c['ncls'], c['K'], c['channels'], max_pooling_flag=c['max_pool'], blind_chunks=4)
        feats = torch.randn((1, hot.fh, hot.fw, c['channels']), device='cuda', generator=gen)
    assert hot.N == want['num_anchors']
    _, _, labels, _, count = hot.step(g(inp['rpn_logits']), g(inp['rpn_deltas']), feats, g(inp['cls_scores']),
                                      g(inp['cls_deltas']))
    torch.cuda.synchronize()
    assert int(hot.nms_done.item()) == 1
    k = int(hot.roi_count.item())
    got = dict(kept_anchor_idx=h(hot.roi_idx[:k]), det_labels=h(labels[:int(count.item())]))
    if c['kind'] == 'fpn':
        got['roi_level'] = h(hot.roi_level[:k]) + 2
        got['level_perm'] = h(hot.roi_perm[:k])
    assert fs.digests(got) == {k_: v for k_, v in want.items() if k_ != 'num_anchors'}


@pytest.mark.parametrize('flag,ch,K', [(True, 64, 300), (False, 128, 100)])
def test_frcnn_step_batch_and_stream_pool_match_single_path(flag, ch, K):
    """Configs 1-2 through the throughput arrangement: B single-level images in the SAME launches
    (odet_fpn_step_t.single_level: FrcnnStepBatch, FrcnnStreamPool) == each image through FrcnnHotPath, bit for bit,
    and == the oracle (base_faster_rcnn_model.py:126-198 minus the dense parts)."""
    from tf_eager_object_detection_amd.pipeline import FrcnnHotPath, FrcnnStepBatch, FrcnnStreamPool, synthetic_frcnn_inputs
    shape, ncls, B = (300, 420), 21, 3
    sets = [synthetic_frcnn_inputs(shape, ncls, K, ch, seed=900 + i) for i in range(B)]
    sb = FrcnnStepBatch(B, shape, ncls, K, ch, max_pooling_flag=flag, blind_chunks=3)
    for b, (_, d) in enumerate(sets):
        sb.bind(b, d['rpn_logits'], d['rpn_deltas'], d['feat'], d['cls_scores'], d['cls_deltas'])
    sb.enqueue(7, B)
    torch.cuda.synchronize()
    ref = FrcnnHotPath(shape, ncls, K, ch, max_pooling_flag=flag, blind_chunks=4)
    from tf_eager_object_detection_amd.utils.anchor_generator import generate_anchor_base
    base = generate_anchor_base(16, [0.5, 1, 2], np.array([8, 16, 32])).astype(np.float32)
    anchors = on.generate_by_anchor_base_tf(base, 16, ref.fh, ref.fw)
    for b, (host, d) in enumerate(sets):
        ref.step(d['rpn_logits'], d['rpn_deltas'], d['feat'], d['cls_scores'], d['cls_deltas'])
        torch.cuda.synchronize()
        s = sb.slots[b]
        assert int(s.nms_done.item()) == 1 and int(ref.nms_done.item()) == 1
        k = int(ref.roi_count.item())
        assert int(s.roi_count.item()) == k and k > 0
        assert torch.equal(s.roi_idx[:k], ref.roi_idx[:k]) and torch.equal(s.rois[:k], ref.rois[:k])
        assert torch.equal(s.roi_features, ref.roi_features)
        assert torch.equal(s.record, ref.record)
        order = h(s.roi_order)
        assert sorted(order.tolist()) == list(range(K))                  # the fused processing order is a permutation
        fg = co.rpn_fg_frcnn(host['rpn_logits'], 9)
        want_rois, want_idx = co.region_proposal(host['rpn_deltas'], anchors, fg, shape, K, 0.7)
        np.testing.assert_array_equal(h(s.roi_idx[:k]), want_idx)
        np.testing.assert_array_equal(h(s.roi_features[:k]), co.roi_pool(host['feat'][0], want_rois, stride=16, pool=7, max_pool=flag))
    pool = FrcnnStreamPool(1, shape, ncls, K, ch, batch=B, max_pooling_flag=flag, blind_chunks=3)
    try:
        for b, (_, d) in enumerate(sets):
            pool.bind(b, d['rpn_logits'], d['rpn_deltas'], d['feat'], d['cls_scores'], d['cls_deltas'])
        pool.submit_group(0)
        pool.wait()
        torch.cuda.synchronize()
        for b in range(B):
            assert torch.equal(pool.slots[b].record, sb.slots[b].record)
            assert torch.equal(pool.slots[b].roi_features, sb.slots[b].roi_features)
    finally:
        pool.close()


def test_zz_report_ulp_histograms(capsys):
    """(last test of the file) the distances the parity assertions above actually saw, per output kind"""
    with capsys.disabled():
        print()
        for k in sorted(ULP_SEEN):
            hgram = ULP_SEEN[k]
            print('  ulp distance %-36s 0: %-10d 1: %-8d 2: %-6d 3: %-4d 4+: %d' % ((k,) + tuple(int(v) for v in hgram)))


def test_nms_tie_split_of_an_overflowing_boundary_bin():
    """A saturated RPN puts tens of thousands of EQUAL scores at the top of the order (softmax == 1.0f): the selection's
    boundary bin does not fit and is split exactly in (score desc, index asc) order (k_sel_tie_hist / k_sel_tie_take), so
    the sync-free chunks from the selection finish the job instead of the full sort -- kept indices equal the oracle's."""
    rng = np.random.default_rng(23)
    done = torch.zeros(1, dtype=torch.int32, device='cuda')
    n = 60000
    boxes = syn.random_boxes(n, (800, 1333), rng, 24, 300)
    # (a) one plateau of 25000 ties at the very top, distinct scores below it
    scores = syn.scores_distinct(n, rng) * np.float32(0.9)
    tied = rng.permutation(n)[:25000]
    scores[tied] = np.float32(1.0)
    want, stats = co.nms(boxes, scores, 1000, 0.7, True)
    assert stats[0] < 8192                                     # the oracle needs fewer candidates than the selection holds
    for blind in (2, 3):
        done.zero_()
        idx, cnt = ops.nms(g(boxes), g(scores), 1000, 0.7, blind_chunks=blind, done=done)
        assert int(done.item()) == 1
        np.testing.assert_array_equal(h(idx[:int(cnt.item())]), want)
    # (b) plateaus of neighbouring float32 values (the boundary value falls inside the bin's last byte) + a tied boundary
    scores = syn.scores_distinct(n, rng) * np.float32(0.5)
    vals = np.float32(1.0) - np.arange(40, dtype=np.float32) * np.float32(2 ** -24)      # 1.0, 1 - 1 ulp, ...
    scores[tied] = vals[rng.integers(0, 40, tied.shape[0])]
    want, stats = co.nms(boxes, scores, 1000, 0.7, True)
    done.zero_()
    idx, cnt = ops.nms(g(boxes), g(scores), 1000, 0.7, blind_chunks=2, done=done)
    assert int(done.item()) == 1
    np.testing.assert_array_equal(h(idx[:int(cnt.item())]), want)
    # (c) two dense clusters, every score tied: the split candidates keep 2 boxes, the full order has to be walked to its
    #     end (chunk 0 + chunk 1 from the split selection, then 14 chunks of the full order: exact, prefix-consistent)
    base = np.float32([100, 100, 300, 260])
    cl = base + rng.uniform(-3, 3, (n, 4)).astype(np.float32)
    cl[::400] += np.float32([500, 300, 500, 300])
    sc = np.full(n, 1.0, np.float32)
    want = co.nms(cl, sc, 100, 0.5)
    done.zero_()
    idx, cnt = ops.nms(g(cl), g(sc), 100, 0.5, blind_chunks=17, done=done)
    assert int(done.item()) == 1
    np.testing.assert_array_equal(h(idx[:int(cnt.item())]), want)
    # (d) the FPN proposal stage with saturated logits, batched launches (FpnStepBatch), against the oracle per image
    from tf_eager_object_detection_amd.pipeline import FpnStepBatch, synthetic_fpn_inputs
    shape, K = (320, 480), 300
    sb = FpnStepBatch(3, shape, 21, K, 16, blind_chunks=2)
    hosts = []
    for b in range(3):
        host, dev = synthetic_fpn_inputs(shape, 21, K, 16, seed=40 + b)
        lg = host['rpn_logits'].copy()
        sat = np.random.default_rng(b).permutation(lg.shape[0])[:lg.shape[0] // 3]
        lg[sat, 1] = lg[sat, 0] + np.float32(40.0)             # fg - bg = 40: softmax == 1.0f exactly
        host['rpn_logits'] = lg
        hosts.append(host)
        sb.bind(b, g(lg), dev['rpn_deltas'], dev['feats'], dev['cls_scores'], dev['cls_deltas'])
    sb.enqueue(sb.STAGE_PROPOSALS, 3)
    torch.cuda.synchronize()
    anchors = co.fpn_anchors(shape)
    for b in range(3):
        fg = co.rpn_fg_fpn(hosts[b]['rpn_logits'])
        assert (fg == 1.0).sum() > 8192
        rois, widx = co.region_proposal(hosts[b]['rpn_deltas'], anchors, fg, shape, K, 0.7)
        hs = sb.slots[b]
        assert int(hs.nms_done.item()) == 1
        k = int(hs.roi_count.item())
        np.testing.assert_array_equal(h(hs.roi_idx[:k]), widx)


def test_step_batch_beyond_eight_images_and_direct_group_enqueue():
    """FpnStepBatch with more than 8 images goes out as consecutive 8-image launch sequences (round 3: 15 / 30 images per
    detector pass) -- bit-identical, image by image, to the single-image path; FpnStreamPool.enqueue_group (the multi-rank
    loop's enqueue by the calling thread) gives the same records as submit_group through the executor thread."""
    from tf_eager_object_detection_amd.pipeline import FpnHotPath, FpnStepBatch, FpnStreamPool, synthetic_fpn_inputs
    shape, K, C, B = (224, 320), 200, 32, 11
    sb = FpnStepBatch(B, shape, 21, K, C, blind_chunks=2)
    devs = []
    for b in range(B):
        _, dev = synthetic_fpn_inputs(shape, 21, K, C, seed=300 + b, score_kind='clustered' if b % 3 == 0 else 'distinct')
        devs.append(dev)
        sb.bind(b, dev['rpn_logits'], dev['rpn_deltas'], dev['feats'], dev['cls_scores'], dev['cls_deltas'])
    sb.enqueue(7, B)
    torch.cuda.synchronize()
    assert sb.nms_done_all.tolist() == [1] * B
    ref = FpnHotPath(shape, 21, K, C, blind_chunks=3)
    for b in range(B):
        d = devs[b]
        feats, boxes, labels, scores, count = ref.step(d['rpn_logits'], d['rpn_deltas'], d['feats'], d['cls_scores'], d['cls_deltas'])
        torch.cuda.synchronize()
        hs = sb.slots[b]
        k = int(ref.roi_count.item())
        assert k == int(hs.roi_count.item())
        assert torch.equal(ref.roi_idx[:k], hs.roi_idx[:k]) and torch.equal(feats[:k], sb.roi_features[b][:k])
        m = int(count.item())
        assert m == int(hs.det_count.item()) and torch.equal(boxes[:m], hs.det_boxes[:m])
        assert torch.equal(labels[:m], hs.det_labels[:m]) and torch.equal(scores[:m], hs.det_scores[:m])
    # the two ways of enqueuing a stream group
    pool = FpnStreamPool(2, shape, 21, K, C, batch=4, blind_chunks=2)
    for k_ in range(pool.n):
        d = devs[k_]
        pool.bind(k_, d['rpn_logits'], d['rpn_deltas'], d['feats'], d['cls_scores'], d['cls_deltas'])
    for g_ in range(2):
        pool.submit_group(g_)
    pool.wait()
    torch.cuda.synchronize()
    want = [s.record.clone() for s in pool.slots]
    for s in pool.slots:
        s.record.zero_()
    for g_ in range(2):
        pool.enqueue_group(g_)
    torch.cuda.synchronize()
    for s, w in zip(pool.slots, want):
        assert torch.equal(s.record, w) and float(w[-1]) > 0
    pool.close()

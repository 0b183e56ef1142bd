"""ResNet-FPN detector counterpart: shape bookkeeping and the TF1 legacy bilinear
resize on CPU; on the GPU the assembled model's hot-path state against the oracle, given the tensors the
dense parts produced."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import c_oracle as co
from tf_eager_object_detection_amd import synthetic as syn

import torch_reference as tref


def _np_legacy_resize(x, oh, ow):
    """tf.image.resize_bilinear (TF 1.x, align_corners=False) restated with numpy loops; x [H,W]."""
    H, W = x.shape
    out = np.zeros((oh, ow), np.float32)
    for y in range(oh):
        iy = np.float32(y) * np.float32(H / oh)
        t = int(np.floor(iy)); b = min(t + 1, H - 1); wy = np.float32(iy - t)
        for xx in range(ow):
            ix = np.float32(xx) * np.float32(W / ow)
            l = int(np.floor(ix)); r = min(l + 1, W - 1); wx = np.float32(ix - l)
            top = x[t, l] + (x[t, r] - x[t, l]) * wx
            bot = x[b, l] + (x[b, r] - x[b, l]) * wx
            out[y, xx] = top + (bot - top) * wy
    return out


def test_tf_legacy_resize_bilinear_cpu():
    from tf_eager_object_detection_amd.model.fpn_detector import tf_legacy_resize_bilinear
    rng = np.random.default_rng(0)
    for (h, w), (oh, ow) in (((13, 21), (25, 42)), ((25, 42), (50, 84)), ((5, 7), (9, 13))):
        x = rng.standard_normal((h, w)).astype(np.float32)
        got = tf_legacy_resize_bilinear(torch.from_numpy(x)[None, None], (oh, ow))[0, 0].numpy()
        np.testing.assert_allclose(got, _np_legacy_resize(x, oh, ow), rtol=0, atol=1e-6)


def test_oracle_topdown_merge_matches_loop_restatement_cpu():
    """oracle_np.tf_resize_bilinear_legacy / fpn_topdown_merge against the scalar loop restatement above."""
    from oracle import oracle_np as on
    rng = np.random.default_rng(3)
    for (h, w), (oh, ow) in (((13, 21), (25, 42)), ((25, 42), (50, 84)), ((5, 7), (9, 13)), ((6, 6), (6, 6))):
        top = rng.standard_normal((2, h, w, 3)).astype(np.float32)
        lat = rng.standard_normal((2, oh, ow, 3)).astype(np.float32)
        up = on.tf_resize_bilinear_legacy(top, (oh, ow))
        for b in range(2):
            for c in range(3):
                np.testing.assert_array_equal(up[b, :, :, c], _np_legacy_resize(top[b, :, :, c], oh, ow))
        want = (up * np.float32(0.5) + lat * np.float32(0.5)).astype(np.float32)
        np.testing.assert_array_equal(on.fpn_topdown_merge(top, lat), want)


@pytest.mark.gpu
@pytest.mark.parametrize('hw,HW,C', [((25, 42), (50, 84), 256), ((50, 84), (100, 167), 256), ((7, 5), (13, 10), 8),
                                      ((9, 9), (9, 9), 16)])
def test_fpn_topdown_merge_bit_exact(hw, HW, C):
    """odet_fpn_topdown_merge (float32): identical bits to the TF1 restatement; float16 maps: float32 arithmetic
    on the float16 inputs, one rounding at the end."""
    from oracle import oracle_np as on
    from tf_eager_object_detection_amd import ops
    rng = np.random.default_rng(C + hw[0])
    B = 3
    top = rng.standard_normal((B,) + hw + (C,)).astype(np.float32)
    lat = rng.standard_normal((B,) + HW + (C,)).astype(np.float32)
    got = ops.fpn_topdown_merge(torch.from_numpy(top).cuda(), torch.from_numpy(lat).cuda()).cpu().numpy()
    np.testing.assert_array_equal(got, on.fpn_topdown_merge(top, lat))
    t16, l16 = top.astype(np.float16), lat.astype(np.float16)
    got16 = ops.fpn_topdown_merge(torch.from_numpy(t16).cuda(), torch.from_numpy(l16).cuda()).cpu().numpy()
    assert got16.dtype == np.float16
    want16 = on.fpn_topdown_merge(t16.astype(np.float32), l16.astype(np.float32)).astype(np.float16)
    np.testing.assert_array_equal(got16, want16)
    with pytest.raises(TypeError):
        ops.fpn_topdown_merge(torch.from_numpy(top).cuda(), torch.from_numpy(l16).cuda())


@pytest.mark.gpu
def test_detector_neck_uses_fused_merge_and_matches_torch_formulation():
    from tf_eager_object_detection_amd.model.fpn_detector import ResNetFpnDetector, tf_legacy_resize_bilinear
    torch.manual_seed(2)
    cl = torch.channels_last
    top = torch.randn(2, 256, 13, 21, device='cuda').contiguous(memory_format=cl)
    lat = torch.randn(2, 256, 25, 42, device='cuda').contiguous(memory_format=cl)
    got = ResNetFpnDetector._merge(top, lat)
    want = tf_legacy_resize_bilinear(top, (25, 42)) * 0.5 + lat * 0.5
    assert got.shape == want.shape and got.is_contiguous(memory_format=cl)
    torch.testing.assert_close(got, want, rtol=0, atol=2e-6)


@pytest.mark.gpu
@pytest.mark.parametrize('C', [8, 64, 256])
def test_bias_act_epilogue(C):
    """odet_bias_act: float32 identical to the separate framework ops ((x + b) + r, then relu, NaN kept);
    float16 = float32 arithmetic on the float16 inputs with one rounding."""
    from tf_eager_object_detection_amd import ops
    g = torch.Generator(device='cuda'); g.manual_seed(C)
    x = torch.randn(3, 17, 19, C, device='cuda', generator=g)
    b = torch.randn(C, device='cuda', generator=g)
    r = torch.randn(3, 17, 19, C, device='cuda', generator=g)
    x[0, 0, 0, 0] = float('nan')
    for res in (None, r):
        for relu in (False, True):
            want = x + b
            if res is not None:
                want = want + res
            if relu:
                want = torch.relu(want)
            got = ops.bias_act_(x.clone(), b, res, relu)
            assert torch.equal(torch.nan_to_num(got, nan=123.0), torch.nan_to_num(want, nan=123.0))
            x16, b16 = x.half(), b.half()
            r16 = None if res is None else res.half()
            w = x16.float() + b16.float()
            if r16 is not None:
                w = w + r16.float()
            if relu:
                w = torch.relu(w)
            got16 = ops.bias_act_(x16.clone(), b16, r16, relu)
            assert torch.equal(torch.nan_to_num(got16, nan=123.0), torch.nan_to_num(w.half(), nan=123.0))
    with pytest.raises(TypeError):
        ops.bias_act_(x.clone(), b.half())


@pytest.mark.gpu
def test_detector_block_with_fused_epilogue_matches_torch_formulation():
    from tf_eager_object_detection_amd.model import fpn_detector as fd
    torch.manual_seed(5)
    blk = fd._Block(64, 64, 2, True).cuda().to(memory_format=torch.channels_last).eval()
    with torch.no_grad():
        for m in (blk.short, blk.c1, blk.c2, blk.c3):
            m.bias.normal_(0, 0.1)
        x = torch.randn(2, 64, 24, 30, device='cuda').contiguous(memory_format=torch.channels_last)
        got = blk(x)
        sc = blk.short(x)
        y = torch.relu(blk.c1(x)); y = torch.relu(blk.c2(y)); y = blk.c3(y)
        want = torch.relu(sc + y)
    assert got.shape == want.shape
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5)


def test_feature_map_sizes_match_anchor_grids_cpu():
    """conv arithmetic of the extractor/neck must give ceil(H/stride) x ceil(W/stride)."""
    from tf_eager_object_detection_amd.model.fpn_detector import ResNetFpnDetector
    m = ResNetFpnDetector(50, 21, (160, 224), 100, dtype=torch.float32).eval()
    img = torch.zeros((1, 160, 224, 3))
    with torch.no_grad():
        ps = tref.fpn_features(m, img)             # (the product has no CPU route: the plain-torch formulation of the same modules)
    want = syn.fpn_level_shapes((160, 224))
    assert [tuple(p.shape[2:]) for p in ps] == [tuple(s) for s in want]
    with torch.no_grad():
        sc, dl = tref.fpn_rpn(m, ps)
    assert sc.shape == (1, syn.num_fpn_anchors((160, 224)), 2) and dl.shape[-1] == 4


@pytest.mark.gpu
def test_detector_hot_path_state_matches_oracle():
    from tf_eager_object_detection_amd.model.fpn_detector import ResNetFpnDetector
    torch.manual_seed(1)
    shape = (256, 352)
    m = ResNetFpnDetector(50, 21, shape, 300, dtype=torch.float32, blind_chunks=3).prepare()
    rng = np.random.default_rng(1)
    img = torch.from_numpy((rng.uniform(0, 255, (1,) + shape + (3,)) - 110).astype(np.float32)).cuda()
    out = m(img)
    torch.cuda.synchronize()
    hot = m._hot[0]
    assert int(hot.nms_done.item()) == 1
    with torch.no_grad():
        ps = m.features(img)
        sc, dl = m.rpn(ps)
        hot.stage_proposals(sc[0].float().contiguous(), dl[0].float().contiguous())
        hot.stage_roi([p.permute(0, 2, 3, 1).float().contiguous() for p in ps[:4]])
    torch.cuda.synchronize()
    logits = sc[0].float().cpu().numpy()
    deltas = dl[0].float().cpu().numpy()
    fg = co.rpn_fg_fpn(logits)
    rois, idx = co.region_proposal(deltas, co.fpn_anchors(shape), fg, shape, 300, 0.7)
    k = int(hot.roi_count.item())
    assert k == len(idx)
    np.testing.assert_array_equal(hot.roi_idx[:k].cpu().numpy(), idx)
    lv, perm, cnt = co.assign_levels(rois)
    np.testing.assert_array_equal(hot.roi_perm[:k].cpu().numpy(), perm)
    srois = rois[perm]
    maps = [p.permute(0, 2, 3, 1).float().cpu().numpy()[0] for p in ps[:4]]
    want = np.concatenate([co.roi_pool(maps[l], srois[lv[perm] == l + 2], image_shape=shape, pool=7)
                           for l in range(4) if np.any(lv == l + 2)], axis=0)
    got = hot.roi_features[:k].cpu().numpy()
    assert np.max(np.abs(got - want)) <= 1e-4 * max(1.0, float(np.abs(want).max()))
    boxes, labels, scores, count = out[0]
    assert int(count.item()) > 0


@pytest.mark.gpu
def test_detector_hip_graph_replay_matches_eager():
    """ResNetFpnDetector.capture: the whole forward pass replayed as one HIP graph gives the eager outputs,
    and follows its input (a second image through the same graph)."""
    from tf_eager_object_detection_amd.model.fpn_detector import ResNetFpnDetector
    torch.manual_seed(3)
    shape = (256, 352)
    m = ResNetFpnDetector(50, 21, shape, 300, dtype=torch.float32, blind_chunks=3).prepare()
    rng = np.random.default_rng(7)
    imgs = [torch.from_numpy((rng.uniform(0, 255, (1,) + shape + (3,)) - 110).astype(np.float32)).cuda() for _ in range(2)]
    eager = []
    for im in imgs:
        b, l, s, c = m(im)[0]
        eager.append((b.clone(), l.clone(), s.clone(), int(c.item())))
    run = m.capture(1)
    for im, (eb, el, es, ec) in zip(imgs, eager):
        b, l, s, c = run(im)[0]
        torch.cuda.synchronize()
        assert int(c.item()) == ec and ec > 0
        assert torch.equal(l[:ec], el[:ec])
        # (library convolutions may pick another algorithm during the capture warm-up: low-bit differences)
        torch.testing.assert_close(b[:ec], eb[:ec], rtol=1e-3, atol=1e-2)
        torch.testing.assert_close(s[:ec], es[:ec], rtol=1e-3, atol=1e-4)
    assert int(m._hot[0].nms_done.item()) == 1


def test_c4_feature_map_size_matches_anchor_grid_cpu():
    from tf_eager_object_detection_amd.model.frcnn_detector import ResNetC4Detector
    m = ResNetC4Detector(50, 21, (160, 224), 50, dtype=torch.float32).eval()
    with torch.no_grad():
        c4 = tref.c4_features(m, torch.zeros((1, 160, 224, 3)))
        sc, dl = tref.frcnn_rpn(m, c4)
    assert tuple(c4.shape) == (1, 1024, 10, 14)
    assert sc.shape == (1, 10 * 14, 18) and dl.shape == (1, 10 * 14 * 9, 4)


@pytest.mark.gpu
def test_c4_detector_hot_path_state_matches_oracle():
    """ResNet-50 C4 Faster R-CNN (BASELINE config 2): given the tensors the dense parts produced, the hot-path
    state (kept anchors, RoIs, 14x14 + max-pooled features) matches the oracle."""
    from tf_eager_object_detection_amd.model.frcnn_detector import ResNetC4Detector
    from oracle import oracle_np as on
    torch.manual_seed(4)
    shape = (256, 352)
    m = ResNetC4Detector(50, 21, shape, 100, dtype=torch.float32, blind_chunks=4).prepare()
    rng = np.random.default_rng(4)
    img = torch.from_numpy((rng.uniform(0, 255, (1,) + shape + (3,)) - 110).astype(np.float32)).cuda()
    out = m(img)
    torch.cuda.synchronize()
    hot = m._hot[0]
    assert int(hot.nms_done.item()) == 1
    # (library convolutions are not bit-reproducible from call to call: the hot path is re-run on the very
    # tensors the oracle gets)
    with torch.no_grad():
        c4 = m.features(img)
        sc, dl = m.rpn(c4)
        hot.stage_proposals(sc[0].float().contiguous(), dl[0].float().contiguous())
        hot.stage_roi(c4.permute(0, 2, 3, 1).float().contiguous())
    torch.cuda.synchronize()
    fg = co.rpn_fg_frcnn(sc[0].float().cpu().numpy(), 9)
    anchors = co.anchors_shift(hot.anchor_base, 16, hot.fh, hot.fw)
    rois, idx = co.region_proposal(dl[0].float().cpu().numpy(), anchors, fg, shape, 100, 0.7)
    k = int(hot.roi_count.item())
    assert k == len(idx)
    np.testing.assert_array_equal(hot.roi_idx[:k].cpu().numpy(), idx)
    fmap = c4.permute(0, 2, 3, 1).float().cpu().numpy()[0]
    want = co.roi_pool(fmap, rois, stride=16, pool=7, max_pool=False)       # resnet_roi_pooling_max_pooling_flag: False
    got = hot.roi_features[:k].cpu().numpy()
    assert np.max(np.abs(got - want)) <= 1e-4 * max(1.0, float(np.abs(want).max()))
    boxes, labels, scores, count = out[0]
    assert 0 < int(count.item()) <= 50


def test_vgg16_feature_map_size_matches_anchor_grid_cpu():
    from tf_eager_object_detection_amd.model.frcnn_detector import Vgg16Detector
    m = Vgg16Detector(21, (150, 200), 50, dtype=torch.float32).eval()
    with torch.no_grad():
        f = tref.vgg16_features(m, torch.zeros((1, 150, 200, 3)))
        sc, dl = tref.frcnn_rpn(m, f)
    assert tuple(f.shape) == (1, 512, 10, 13)                      # ceil(150/16) x ceil(200/16)
    assert sc.shape == (1, 130, 18) and dl.shape == (1, 130 * 9, 4)


@pytest.mark.gpu
def test_vgg16_detector_runs_and_pools_like_oracle():
    """VGG16 Faster R-CNN (BASELINE config 1 shapes, reduced): RoIs and 14x14 + max pooled features vs oracle."""
    from tf_eager_object_detection_amd.model.frcnn_detector import Vgg16Detector
    torch.manual_seed(6)
    shape = (240, 320)
    m = Vgg16Detector(21, shape, 100, dtype=torch.float32, blind_chunks=4).prepare()
    rng = np.random.default_rng(6)
    img = torch.from_numpy((rng.uniform(0, 255, (1,) + shape + (3,)) - 110).astype(np.float32)).cuda()
    out = m(img)
    torch.cuda.synchronize()
    hot = m._hot[0]
    assert int(hot.nms_done.item()) == 1
    with torch.no_grad():
        f = m.features(img)
        sc, dl = m.rpn(f)
        hot.stage_proposals(sc[0].float().contiguous(), dl[0].float().contiguous())
        hot.stage_roi(f.permute(0, 2, 3, 1).float().contiguous())
    torch.cuda.synchronize()
    fg = co.rpn_fg_frcnn(sc[0].float().cpu().numpy(), 9)
    anchors = co.anchors_shift(hot.anchor_base, 16, hot.fh, hot.fw)
    rois, idx = co.region_proposal(dl[0].float().cpu().numpy(), anchors, fg, shape, 100, 0.7)
    k = int(hot.roi_count.item())
    np.testing.assert_array_equal(hot.roi_idx[:k].cpu().numpy(), idx)
    fmap = f.permute(0, 2, 3, 1).float().cpu().numpy()[0]
    want = co.roi_pool(fmap, rois, stride=16, pool=7, max_pool=True)
    got = hot.roi_features[:k].cpu().numpy()
    assert np.max(np.abs(got - want)) <= 1e-4 * max(1.0, float(np.abs(want).max()))
    assert 0 < int(out[0][3].item()) <= 50


@pytest.mark.gpu
def test_detector_batched_hot_path_matches_per_image_path():
    """ResNetFpnDetector (default arrangement): the images of a batch share the hot-path launches (FpnStepBatch)
    and the RoI head runs on all crops at once -- same detections as the per-image path on the same dense
    outputs."""
    from tf_eager_object_detection_amd.model.fpn_detector import ResNetFpnDetector
    torch.manual_seed(8)
    shape, K, B = (256, 352), 300, 3
    m = ResNetFpnDetector(50, 21, shape, K, dtype=torch.float32, max_batch=B, blind_chunks=2).prepare()
    assert m._steps is not None
    rng = np.random.default_rng(8)
    img = torch.from_numpy((rng.uniform(0, 255, (B,) + shape + (3,)) - 110).astype(np.float32)).cuda()
    with torch.no_grad():
        ps = m.features(img)
        sc, dl = m.rpn(ps)
        sc, dl = sc.float().contiguous(), dl.float().contiguous()
        maps = [p.permute(0, 2, 3, 1).float().contiguous() for p in ps[:4]]
        got = m._forward_batched(B, sc, dl, maps)
        got = [tuple(t.clone() for t in g) for g in got]
        torch.cuda.synchronize()
        assert all(int(h.nms_done.item()) == 1 for h in m._hot)
        from tf_eager_object_detection_amd.pipeline import FpnHotPath
        ref = FpnHotPath(shape, 21, K, 256, blind_chunks=3)
        for b in range(B):
            ref.stage_proposals(sc[b], dl[b])
            feats = ref.stage_roi([mm[b:b + 1] for mm in maps])
            logits, bbox = m.roi_head(feats)
            cls = torch.softmax(logits.float(), dim=-1).contiguous()
            boxes, labels, scores, count = ref.stage_detect(cls, bbox.float().contiguous())
            torch.cuda.synchronize()
            c = int(count.item())
            assert c == int(got[b][3].item()) and c > 0
            assert torch.equal(labels[:c], got[b][1][:c])
            torch.testing.assert_close(boxes[:c], got[b][0][:c], rtol=1e-4, atol=1e-3)
            torch.testing.assert_close(scores[:c], got[b][2][:c], rtol=1e-4, atol=1e-5)
    # and the public forward takes the same route
    out = m(img)
    torch.cuda.synchronize()
    assert len(out) == B and all(int(o[3].item()) > 0 for o in out)


@pytest.mark.gpu
def test_rpn_pack_and_detector_rpn_match_torch_formulation():
    """odet_rpn_pack: bias + float32 + the level's slice of the concatenated arrays (base_fpn_model.py:188-200,
    427-432) -- exact against reshape / concat of the biased convolution outputs."""
    from tf_eager_object_detection_amd import ops
    from tf_eager_object_detection_amd.model.fpn_detector import ResNetFpnDetector
    g = torch.Generator(device='cuda'); g.manual_seed(9)
    B, A = 2, 3
    shapes = [(12, 17), (6, 9), (3, 5)]
    n = sum(h * w for h, w in shapes) * A
    for dt in (torch.float32, torch.float16):
        outs = torch.zeros((B, n, 2), dtype=torch.float32, device='cuda')
        outd = torch.zeros((B, n, 4), dtype=torch.float32, device='cuda')
        want_s, want_d, pairs, off = [], [], [], 0
        bs = torch.randn(2 * A, device='cuda', generator=g).to(dt)
        bd = torch.randn(4 * A, device='cuda', generator=g).to(dt)
        for h, w in shapes:
            s = torch.randn(B, h, w, 2 * A, device='cuda', generator=g).to(dt)
            d = torch.randn(B, h, w, 4 * A, device='cuda', generator=g).to(dt)
            pairs.append(torch.cat([s, d], 3).contiguous())
            ops.rpn_pack(s, bs, outs, off * 2)
            ops.rpn_pack(d, bd, outd, off * 4)
            want_s.append((s.float() + bs.float()).reshape(B, -1, 2))
            want_d.append((d.float() + bd.float()).reshape(B, -1, 4))
            off += h * w * A
        assert torch.equal(outs, torch.cat(want_s, 1)) and torch.equal(outd, torch.cat(want_d, 1))
        # the two convolutions as one contraction: [B,h,w,6A] = scores then deltas along the channel
        outs2, outd2, off = torch.zeros_like(outs), torch.zeros_like(outd), 0
        for (h, w), sd in zip(shapes, pairs):
            ops.rpn_pack_pair(sd, torch.cat([bs, bd]), A, outs2, outd2, off)
            off += h * w * A
        assert torch.equal(outs2, outs) and torch.equal(outd2, outd)
    with pytest.raises(Exception):
        ops.rpn_pack_pair(torch.zeros(1, 2, 2, 18, device='cuda'), torch.zeros(18, device='cuda'), 3,
                          torch.zeros(1, 10, 2, device='cuda'), torch.zeros(1, 12, 4, device='cuda'), 0)   # 12 anchors, 10 slots
    with pytest.raises(Exception):
        ops.rpn_pack(torch.zeros(1, 2, 2, 6, device='cuda'), torch.zeros(6, device='cuda'),
                     torch.zeros(1, 10, 2, device='cuda'), 0)                       # 24 values do not fit 20
    torch.manual_seed(10)
    m = ResNetFpnDetector(50, 21, (128, 160), 50, dtype=torch.float32).prepare()
    with torch.no_grad():
        for c in (m.rpn_score, m.rpn_bbox):
            c.bias.normal_(0, 0.1)
        ps = [torch.randn(2, 256, h, w, device='cuda').contiguous(memory_format=torch.channels_last)
              for h, w in ((32, 40), (16, 20), (8, 10), (4, 5), (2, 3))]
        sc, dl = m.rpn(ps)
        ws, wd = [], []
        for p in ps:
            x = torch.relu(m.rpn_conv(p))
            ws.append(m.rpn_score(x).permute(0, 2, 3, 1).reshape(2, -1, 2))
            wd.append(m.rpn_bbox(x).permute(0, 2, 3, 1).reshape(2, -1, 4))
    torch.testing.assert_close(sc, torch.cat(ws, 1), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(dl, torch.cat(wd, 1), rtol=1e-5, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize('M,K,N', [(1007, 64, 256), (96, 128, 512), (33, 256, 1024), (2500, 256, 64), (64, 64, 320),
                                   (5000, 128, 128), (777, 512, 2048), (130, 512, 64), (300, 512, 192)])
def test_conv1x1_mfma_kernel_exact_on_integer_data_and_close_on_random(M, K, N):
    """odet_conv1x1_f16 (1x1 convolution + bias + shortcut + ReLU on the matrix cores, resnet_fpn.py:154-205):
    EXACT on small-integer data with asymmetric operands (catches any fragment / permutation / row-column slip),
    within float16 rounding of the float32 formulation on random data; pixel counts that do not fill a tile."""
    from tf_eager_object_detection_amd import ops
    g = torch.Generator(device='cuda'); g.manual_seed(M + K + N)
    ri = lambda lo, hi, *sh: torch.randint(lo, hi, sh, device='cuda', generator=g).to(torch.float16)
    x, w, b, r = ri(-3, 4, M, K), ri(-2, 3, N, K), ri(-8, 9, N), ri(-16, 17, M, N)
    if K == 512:
        x, w = ri(-2, 3, M, K), ri(-1, 2, N, K)                                          # keeps every sum below 2048
    w[:, 0] += torch.arange(N, device='cuda').remainder(5).to(torch.float16)           # every channel distinct-ish
    x[:, 1] += torch.arange(M, device='cuda').remainder(7).to(torch.float16)           # every pixel distinct-ish
    for res, relu in ((r, True), (None, True), (r, False), (None, False)):
        want = x.float() @ w.float().t() + b.float()
        if res is not None:
            want = want + res.float()
        if relu:
            want = torch.relu(want)
        assert float(want.abs().max()) < 2048                                          # exactly representable
        got = ops.conv1x1_f16(x, w.view(N, K, 1, 1), b, res, relu)
        assert torch.equal(got.float(), want)
    x, w, b, r = (torch.randn(s, device='cuda', generator=g).to(torch.float16) for s in ((M, K), (N, K), (N,), (M, N)))
    w = (w.float() / K ** 0.5).to(torch.float16)
    want = torch.relu(x.float() @ w.float().t() + b.float() + r.float())
    got = ops.conv1x1_f16(x, w, b, r, True)
    torch.testing.assert_close(got.float(), want, rtol=2e-3, atol=2e-3)
    # the producer's epilogue on load: x is a convolution without bias / ReLU
    ib = torch.randn(K, device='cuda', generator=g).to(torch.float16)
    xin = torch.relu(x.float() + ib.float()).to(torch.float16)
    assert torch.equal(ops.conv1x1_f16(x, w, b, r, True, in_bias=ib), ops.conv1x1_f16(xin, w, b, r, True))
    # in place on the shortcut, NHWC leading dims
    r2 = r.clone().view(1, M, 1, N)
    out = ops.conv1x1_f16(x.view(1, M, 1, K), w, b, r2, True, out=r2)
    assert out.data_ptr() == r2.data_ptr() and torch.equal(out.view(M, N), got)
    with pytest.raises(Exception):
        ops.conv1x1_f16(x[:, :48].contiguous(), w[:, :48].contiguous(), b, None, True)  # 48 input channels


@pytest.mark.gpu
@pytest.mark.parametrize('A', [3, 4, 1])
def test_rpn_head_tail_mfma_kernel(A):
    """odet_rpn_head_tail_f16 (RpnHead after its 3x3 convolution, base_fpn_model.py:393-434 + the reshape / concat of
    :188-200): exact on integer data (fragment maps, the row order that puts deltas / scores into consecutive
    registers, the level offsets), within float16 rounding of the float32 formulation on random data."""
    from tf_eager_object_detection_amd import ops
    g = torch.Generator(device='cuda'); g.manual_seed(40 + A)
    B, shapes = 2, [(13, 17), (7, 9), (3, 5)]                    # 221 / 63 / 15 pixels: none fills a 128-pixel slab evenly
    n = sum(h * w for h, w in shapes) * A
    for kind in ('int', 'randn'):
        scores = torch.full((B, n, 2), -7.0, dtype=torch.float32, device='cuda')
        deltas = torch.full((B, n, 4), -7.0, dtype=torch.float32, device='cuda')
        if kind == 'int':
            ri = lambda lo, hi, *sh: torch.randint(lo, hi, sh, device='cuda', generator=g).to(torch.float16)
            b1, w, b2 = ri(-2, 3, 512), ri(-2, 3, 6 * A, 512), ri(-8, 9, 6 * A)
            w[:, 0] += torch.arange(6 * A, device='cuda').to(torch.float16)          # every output channel distinct
        else:
            rn = lambda *sh: torch.randn(sh, device='cuda', generator=g).to(torch.float16)
            b1, w, b2 = rn(512), (rn(6 * A, 512).float() / 512 ** 0.5).to(torch.float16), rn(6 * A)
        want_s, want_d, off = [], [], 0
        for h, w_ in shapes:
            c = (torch.randint(-3, 4, (B, h, w_, 512), device='cuda', generator=g).to(torch.float16) if kind == 'int'
                 else torch.randn((B, h, w_, 512), device='cuda', generator=g).to(torch.float16))
            if kind == 'int':
                c[..., 1] += torch.arange(h * w_, device='cuda').remainder(5).view(1, h, w_).to(torch.float16)
            ops.rpn_head_tail(c, b1, w.view(6 * A, 512, 1, 1), b2, A, scores, deltas, off)
            t = torch.relu(c.float() + b1.float()).to(torch.float16).float()
            o = t.reshape(B, -1, 512) @ w.float().t() + b2.float()
            want_s.append(o[..., :2 * A].reshape(B, -1, 2))
            want_d.append(o[..., 2 * A:].reshape(B, -1, 4))
            off += h * w_ * A
        ws, wd = torch.cat(want_s, 1), torch.cat(want_d, 1)
        if kind == 'int':
            assert torch.equal(scores, ws) and torch.equal(deltas, wd)
        else:
            torch.testing.assert_close(scores, ws, rtol=2e-3, atol=2e-3)
            torch.testing.assert_close(deltas, wd, rtol=2e-3, atol=2e-3)
    with pytest.raises(Exception):
        ops.rpn_head_tail(torch.zeros(1, 2, 2, 512, device='cuda', dtype=torch.float16), b1, w, b2, A,
                          torch.zeros(1, 3 * A, 2, device='cuda'), torch.zeros(1, 3 * A, 4, device='cuda'), 0)   # 4 px, 3 fit


@pytest.mark.gpu
@pytest.mark.parametrize('dt', [torch.float16, torch.float32])
def test_bias_relu_maxpool_identical_to_separate_passes(dt):
    """odet_bias_relu_maxpool == max_pool2d(relu(x + bias)) bit for bit: the ResNet stem (resnet_fpn.py:228-259: pad 1 +
    3x3/2) and VGG16's MaxPooling2D((2,2), 2, 'same') (odd sizes: ceil mode)."""
    from tf_eager_object_detection_amd import ops
    g = torch.Generator(device='cuda'); g.manual_seed(3)
    for (B, H, W, C, k, s, pad, ceil) in ((2, 41, 67, 64, 3, 2, 1, False), (1, 400, 667, 64, 3, 2, 1, False),
                                           (2, 75, 100, 128, 2, 2, 0, True), (1, 38, 51, 512, 2, 2, 0, True),
                                           (1, 8, 8, 8, 2, 2, 0, False)):
        x = torch.randn(B, H, W, C, device='cuda', generator=g).to(dt)
        b = torch.randn(C, device='cuda', generator=g).to(dt)
        got = ops.bias_relu_maxpool(x, b, k, s, pad, ceil)
        y = torch.relu((x.float() + b.float())).to(dt) if dt == torch.float16 else torch.relu(x + b)
        want = F.max_pool2d(y.permute(0, 3, 1, 2), k, s, padding=pad, ceil_mode=ceil).permute(0, 2, 3, 1)
        assert got.shape == want.shape, (got.shape, want.shape)
        assert torch.equal(got, want.contiguous())
    with pytest.raises(Exception):
        ops.bias_relu_maxpool(torch.zeros(1, 4, 4, 6, device='cuda'), torch.zeros(6, device='cuda'), 2, 2)     # C % 4


# ---- a18: model-level im_detect + the mAP-producing loop, end to end ---------------------------------------------
def _gt_from_detections(dets, rng, num_classes=21, keep=0.6):
    """synthetic ground truth that makes AP non-trivial: a random subset of the oracle's detections, jittered"""
    boxes, labels = [], []
    for j in range(1, num_classes):
        for row in dets[j]:
            if rng.uniform() < keep:
                jit = rng.normal(0, 0.02, 4).astype(np.float32) * (row[2:4] - row[0:2]).repeat(2)[[0, 2, 1, 3]]
                boxes.append(row[:4] + jit)                  # (2 % of the box size: IoU stays far above 0.5)
                labels.append(j)
    if not boxes:
        return np.zeros((0, 4), np.float32), np.zeros(0, np.int32)
    return np.stack(boxes).astype(np.float32), np.asarray(labels, np.int32)


def _im_detect_eval(model, shapes_scales, seed, rois_attr):
    """detector.im_detect -> pascal_eval.detect_image -> evaluate_detections on the GPU path, and the restated
    reference loop (oracle_np.eval_detect_image, pascal_eval_files_utils.py:76-106) on the SAME dense-part outputs."""
    from oracle import oracle_np as on
    from tf_eager_object_detection_amd.evaluation import pascal_eval as pe
    rng = np.random.default_rng(seed)
    kw = dict(score_threshold=0.0, iou_threshold=0.5, max_objects_per_class=50, max_objects_per_image=50, min_size=10)
    dg, dr, gb, gl = [], [], [], []
    shape = model.image_shape
    for scale in shapes_scales:
        img = torch.from_numpy((rng.uniform(0, 255, (1,) + shape + (3,)) - 110).astype(np.float32)).cuda()
        scores, deltas, rois = model.im_detect(img, scale)[0]
        hot = model._hot[0]
        k = int(hot.roi_count.item())
        assert scores.shape == (k, 21) and deltas.shape == (k, 84) and rois.shape == (k, 4) and k > 0
        raw = getattr(hot, rois_attr)[:k]
        # rois / tf.to_float(img_scale) (base_fpn_model.py:390): a true float32 division
        np.testing.assert_array_equal(rois.cpu().numpy(), (raw.cpu().numpy() / np.float32(scale)).astype(np.float32))
        np.testing.assert_allclose(scores.sum(dim=1).cpu().numpy(), 1.0, rtol=0, atol=1e-5)
        raw_h, raw_w = int(round(shape[0] / scale)), int(round(shape[1] / scale))
        dg.append(pe.detect_image(scores, deltas, rois, 1.0, raw_h, raw_w, **kw))        # (already divided)
        dr.append(on.eval_detect_image(scores.cpu().numpy(), deltas.cpu().numpy(), raw.cpu().numpy(), scale, raw_h,
                                       raw_w, **kw))
        b, l = _gt_from_detections(dr[-1], rng)
        gb.append(b)
        gl.append(l)
    for a, b in zip(dg, dr):
        for j in range(1, 21):
            assert a[j].shape == b[j].shape
            np.testing.assert_array_equal(a[j], b[j])                       # identical rows, class by class
    m_gpu, aps_gpu = pe.evaluate_detections(dg, gb, gl, use_07_metric=True)
    m_ref, aps_ref = pe.evaluate_detections(dr, gb, gl, use_07_metric=True)
    assert abs(m_gpu - m_ref) <= 0.002                                      # BASELINE.json: mAP within +-0.002
    assert aps_gpu == aps_ref
    # the synthetic ground truth is hit (a random-init head puts all its detections into one or two classes; the
    # classes without detections and without ground truth score 0 and pull the MEAN down, so look at the best class)
    assert max(aps_ref) > 0.3
    return m_gpu, m_ref


@pytest.mark.gpu
def test_fpn_im_detect_eval_loop_matches_reference_loop():
    """fp32 ResNet-FPN detector -> im_detect (base_fpn_model.py:364-390) -> the per-image
    evaluation loop -> VOC07 mAP, against the restated reference loop on the same dense-part outputs."""
    from tf_eager_object_detection_amd.model.fpn_detector import ResNetFpnDetector
    torch.manual_seed(11)
    m = ResNetFpnDetector(50, 21, (256, 352), 300, dtype=torch.float32).prepare()
    _im_detect_eval(m, [1.0, 1.6, 0.8, 1.25], 5, 'sorted_rois')
    # level-sorted rows with empty levels dropped (:384-388) == the hot path's sorted RoI list
    hot = m._hot[0]
    k = int(hot.roi_count.item())
    lv = hot.roi_level[:k].cpu().numpy()
    assert np.all(np.diff(lv) >= 0) and int(hot.level_counts.sum().item()) == k


@pytest.mark.gpu
@pytest.mark.parametrize('kind', ['c4', 'vgg16'])
def test_frcnn_im_detect_eval_loop_matches_reference_loop(kind):
    """base_faster_rcnn_model.py:279-306 im_detect on the single-level detectors, same check."""
    from tf_eager_object_detection_amd.model.frcnn_detector import ResNetC4Detector, Vgg16Detector
    torch.manual_seed(12)
    if kind == 'c4':
        m = ResNetC4Detector(50, 21, (256, 352), 100, dtype=torch.float32).prepare()
    else:
        m = Vgg16Detector(21, (256, 352), 100, dtype=torch.float32).prepare()
    _im_detect_eval(m, [1.0, 1.6], 6, 'rois')


@pytest.mark.gpu
def test_detector_incomplete_nms_is_flagged_never_silent():
    """a detector must never run its later stages on a partial / stale RoI list.  With too few sync-free
    NMS chunks for a heavily clustered score map the image is reported EMPTY and flagged (check_complete() raises; forward()
    recovers it: test_detector_recovers_incomplete_nms_in_exact_mode); with the detector's default chunk count the same inputs
    complete and match the oracle."""
    from tf_eager_object_detection_amd.pipeline import FpnHotPath, synthetic_fpn_inputs
    shape, K = (800, 1333), 1000
    host, dev = synthetic_fpn_inputs(shape, 21, K, channels=8, seed=500, score_kind='clustered')
    narrow = FpnHotPath(shape, 21, K, 8, blind_chunks=1)
    narrow.step(dev['rpn_logits'], dev['rpn_deltas'], dev['feats'], dev['cls_scores'], dev['cls_deltas'])
    torch.cuda.synchronize()
    assert int(narrow.nms_done.item()) == 0
    assert int(narrow.roi_count.item()) == 0 and int(narrow.det_count.item()) == 0
    assert float(narrow.roi_features.abs().max().item()) == 0.0
    from tf_eager_object_detection_amd.model.fpn_detector import DEFAULT_BLIND_CHUNKS, _NmsCompleteness

    class Probe(_NmsCompleteness):
        _hot = [narrow]
    with pytest.raises(RuntimeError, match='did not complete'):
        Probe().check_complete(1)
    with pytest.raises(RuntimeError, match='no pass to re-run'):
        Probe()._after_pass(1, None)
    safe = FpnHotPath(shape, 21, K, 8, blind_chunks=DEFAULT_BLIND_CHUNKS)
    safe.step(dev['rpn_logits'], dev['rpn_deltas'], dev['feats'], dev['cls_scores'], dev['cls_deltas'])
    torch.cuda.synchronize()
    assert int(safe.nms_done.item()) == 1
    fg = co.rpn_fg_fpn(host['rpn_logits'])
    _, idx = co.region_proposal(host['rpn_deltas'], co.fpn_anchors(shape), fg, shape, K, 0.7)
    k = int(safe.roi_count.item())
    assert k == len(idx)
    np.testing.assert_array_equal(safe.roi_idx[:k].cpu().numpy(), idx)


@pytest.mark.gpu
@pytest.mark.parametrize('batched', [True, False])
def test_detector_recovers_incomplete_nms_in_exact_mode(batched):
    """the reference's NMS is always exact (model/region_proposal.py:73-81).  An all-tied RPN score map
    (zero score weights: every anchor's fg probability is 0.5, the radix selection cannot split the boundary bin) does not
    complete inside blind_chunks = 1; forward() must not raise and must not report the image empty: only the flagged image
    is re-run in the exact mode and ends with the oracle's proposals and detections; the other image of the batch (normal
    scores through a second detector pass) is untouched."""
    from tf_eager_object_detection_amd.model.fpn_detector import ResNetFpnDetector
    torch.manual_seed(21)
    shape, K = (256, 352), 300
    m = ResNetFpnDetector(50, 21, shape, K, dtype=torch.float32, max_batch=2, blind_chunks=1, batched=batched).prepare()
    with torch.no_grad():
        m.rpn_score.weight.zero_()
        m.rpn_score.bias.zero_()
    rng = np.random.default_rng(5)
    img = torch.from_numpy((rng.uniform(0, 255, (2,) + shape + (3,)) - 110).astype(np.float32)).cuda()
    outs = m(img)                                         # no exception
    torch.cuda.synchronize()
    assert m.nms_reruns == 2 and m.incomplete(2) == []
    sc, dl, maps, heads = m._last_pass
    anchors = co.fpn_anchors(shape)
    for b in range(2):
        hot = m._hot[b]
        logits, deltas = sc[b].cpu().numpy(), dl[b].cpu().numpy()
        assert np.all(logits == 0.0)
        rois, idx = co.region_proposal(deltas, anchors, co.rpn_fg_fpn(logits), shape, K, 0.7)
        k = int(hot.roi_count.item())
        assert k == len(idx) and k > 0
        np.testing.assert_array_equal(hot.roi_idx[:k].cpu().numpy(), idx)
        lv, perm, _ = co.assign_levels(rois)
        np.testing.assert_array_equal(hot.roi_perm[:k].cpu().numpy(), perm)
        cls, dlt = heads[b]
        wb, wl, ws = co.post_ops(cls[:k].cpu().numpy(), dlt[:k].cpu().numpy().reshape(k, -1, 4), rois[perm], shape, [0, 0, 0, 0],
                                 [.1, .1, .2, .2], 50, 50, 0.3, 0.0, 16, 21)
        boxes, labels, scores, count = outs[b]
        n = int(count.item())
        assert n == len(ws) and n > 0
        order, worder = np.lexsort((labels[:n].cpu().numpy(), -scores[:n].cpu().numpy())), np.lexsort((wl, -ws))
        np.testing.assert_array_equal(labels[:n].cpu().numpy()[order], wl[worder])
        np.testing.assert_array_equal(scores[:n].cpu().numpy()[order], ws[worder])
        assert np.max(np.abs(boxes[:n].cpu().numpy()[order] - wb[worder])) <= 1e-4 * max(1.0, float(np.abs(wb).max()))
    # im_detect takes the same route (base_fpn_model.py:364-390): no exception, every proposal of the exact NMS
    det = m.im_detect(img, 1.0)
    assert m.nms_reruns == 4 and all(d[0].shape[0] == int(m._hot[b].roi_count.item()) > 0 for b, d in enumerate(det))


@pytest.mark.gpu
@pytest.mark.parametrize('B,H,W,cin,cout', [(1, 13, 21, 256, 512), (2, 25, 42, 256, 512), (1, 50, 84, 64, 256),
                                             (3, 7, 5, 128, 256), (1, 100, 167, 256, 512), (2, 50, 84, 64, 64),
                                             (1, 100, 167, 128, 128), (2, 31, 45, 64, 192), (1, 200, 334, 64, 64),
                                             (1, 160, 167, 128, 128), (1, 113, 100, 64, 384)])
def test_conv3x3_f16_implicit_gemm(B, H, W, cin, cout):
    """odet_conv3x3_f16 (the RpnHead's 3x3 convolution as a hand-written implicit GEMM on the matrix cores,
    base_fpn_model.py:401-417): EXACT on integer-valued data (every product and partial sum is an integer below 2^24, so
    float32 accumulation order cannot matter), within float16 rounding of a float32 torch convolution on random data;
    zero padding at the borders, pixel tiles that end inside the last row, bias + ReLU epilogue."""
    from tf_eager_object_detection_amd import ops
    g = torch.Generator(device='cuda')
    g.manual_seed(B * 1000 + H)
    x = torch.randint(-3, 4, (B, H, W, cin), device='cuda', generator=g).half()
    w = torch.randint(-2, 3, (cout, cin, 3, 3), device='cuda', generator=g).half().contiguous(memory_format=torch.channels_last)
    got = ops.conv3x3_f16(x, w)
    want = F.conv2d(x.permute(0, 3, 1, 2).float(), w.float(), None, 1, 1).permute(0, 2, 3, 1)
    assert got.shape == (B, H, W, cout)
    assert torch.equal(got.float(), want.half().float())        # integers up to 9 * cin * 6 < 65504: exact in float16 too? no:
    # (values above 2048 are not all representable in float16: compare after the same single rounding)
    xr = (torch.randn((B, H, W, cin), device='cuda', generator=g) * 0.5).half()
    wr = (torch.randn((cout, cin, 3, 3), device='cuda', generator=g) * 0.02).half().contiguous(memory_format=torch.channels_last)
    bias = torch.randn(cout, device='cuda', generator=g).half()
    got = ops.conv3x3_f16(xr, wr, bias, relu=True)
    want = F.relu(F.conv2d(xr.permute(0, 3, 1, 2).float(), wr.float(), bias.float(), 1, 1)).permute(0, 2, 3, 1)
    torch.testing.assert_close(got.float(), want, rtol=2e-3, atol=2e-3)
    # the [cout,3,3,cin] weight form and an explicit output buffer
    out = torch.empty((B, H, W, cout), dtype=torch.float16, device='cuda')
    ops.conv3x3_f16(xr, wr.permute(0, 2, 3, 1).contiguous(), bias, relu=True, out=out)
    assert torch.equal(out, got)


def _conv3x3_tile_height(Ms, tiles_n):
    """mirror of conv3x3_launch's choice of the pixel-tile height (csrc/conv3x3.hip): 16-pixel tiles per wave"""
    best, mt_best = 1e300, 8
    for mt in range(8, 3, -1):
        slabs = sum((M + 32 * mt - 1) // (32 * mt) for M in Ms)
        blocks = (slabs + 7) // 8 * 8 * tiles_n
        cost = ((blocks + 255) // 256) * (mt + 2)
        if cost < best * 0.97:
            best, mt_best = cost, mt
    return mt_best


@pytest.mark.gpu
@pytest.mark.parametrize('H,W,mt', [(20, 84, 4), (328, 100, 5), (209, 200, 6), (300, 167, 7), (349, 167, 8)])
def test_conv3x3_f16_every_tile_height(H, W, mt):
    """the five instantiations of k_conv3x3_f16 (128 / 160 / 192 / 224 / 256-pixel workgroup tiles, picked per launch
    so that the slabs fill whole rounds of the 256 CUs): exact on integer-valued data, the last slab ending inside a
    tile, zero padding at the borders"""
    from tf_eager_object_detection_amd import ops
    assert _conv3x3_tile_height([H * W], 1) == mt
    g = torch.Generator(device='cuda')
    g.manual_seed(H)
    x = torch.randint(-3, 4, (1, H, W, 64), device='cuda', generator=g).half()
    w = torch.randint(-2, 3, (256, 64, 3, 3), device='cuda', generator=g).half().contiguous(memory_format=torch.channels_last)
    got = ops.conv3x3_f16(x, w)
    want = F.conv2d(x.permute(0, 3, 1, 2).float(), w.float(), None, 1, 1).permute(0, 2, 3, 1)
    assert torch.equal(got.float(), want.half().float())


@pytest.mark.gpu
@pytest.mark.parametrize('B,H,W,cin,cout', [(1, 13, 21, 256, 512), (2, 25, 42, 96, 256), (3, 7, 5, 32, 256),
                                             (1, 100, 167, 256, 512)])
def test_conv3x3_f32_implicit_gemm(B, H, W, cin, cout):
    """odet_conv3x3_f32 (the same implicit GEMM on exact-float32 matrix instructions, for the detectors' parity mode):
    EXACT on integer-valued data, within float32 accumulation-order noise of torch's float32 convolution on random
    data (max |difference| <= 2e-5 of the result scale for K = 9 * cin <= 2304 terms); several levels in one launch."""
    from tf_eager_object_detection_amd import ops
    g = torch.Generator(device='cuda')
    g.manual_seed(B * 1000 + H)
    x = torch.randint(-3, 4, (B, H, W, cin), device='cuda', generator=g).float()
    w = torch.randint(-2, 3, (cout, cin, 3, 3), device='cuda', generator=g).float().contiguous(memory_format=torch.channels_last)
    prev = torch.backends.cudnn.allow_tf32
    torch.backends.cudnn.allow_tf32 = False
    try:
        got = ops.conv3x3_f32(x, w)
        want = F.conv2d(x.permute(0, 3, 1, 2), w, None, 1, 1).permute(0, 2, 3, 1)
        assert got.shape == (B, H, W, cout)
        assert torch.equal(got, want)
        xr = torch.randn((B, H, W, cin), device='cuda', generator=g) * 0.5
        wr = (torch.randn((cout, cin, 3, 3), device='cuda', generator=g) * 0.02).contiguous(memory_format=torch.channels_last)
        bias = torch.randn(cout, device='cuda', generator=g)
        got = ops.conv3x3_f32(xr, wr, bias, relu=True)
        want64 = F.relu(F.conv2d(xr.permute(0, 3, 1, 2).double(), wr.double(), bias.double(), 1, 1)).permute(0, 2, 3, 1)
        assert float((got.double() - want64).abs().max().item()) <= 2e-5 * max(1.0, float(want64.abs().max().item()))
        out = torch.empty((B, H, W, cout), dtype=torch.float32, device='cuda')
        ops.conv3x3_f32(xr, wr.permute(0, 2, 3, 1).contiguous(), bias, relu=True, out=out)
        assert torch.equal(out, got)
        # two maps in one launch = the two single launches, bit for bit
        x2 = torch.randn((B, max(1, H // 2), W + 3, cin), device='cuda', generator=g)
        ys = ops.conv3x3_f32_levels([xr, x2], wr, bias, relu=True)
        assert torch.equal(ys[0], got) and torch.equal(ys[1], ops.conv3x3_f32(x2, wr, bias, relu=True))
    finally:
        torch.backends.cudnn.allow_tf32 = prev



@pytest.mark.gpu
def test_fp32_rpn_head_own_conv_matches_library_route():
    """The float32 detector's RpnHead (grouped exact-float32 convolution + exact-float32 GEMM + pack) against the plain-torch
    formulation on the library's convolutions: float32 accumulation-order noise only; and bit-reproducible."""
    from tf_eager_object_detection_amd.model import fpn_detector as fd
    torch.manual_seed(5)
    shape = (256, 352)
    m = fd.ResNetFpnDetector(50, 21, shape, 100, dtype=torch.float32, blind_chunks=3).prepare()
    rng = np.random.default_rng(5)
    img = torch.from_numpy((rng.uniform(0, 255, (1,) + shape + (3,)) - 110).astype(np.float32)).cuda()
    with torch.no_grad():
        ps = m.features(img)
        s_own, d_own = m.rpn(ps)
        s_own2, d_own2 = m.rpn(ps)
        s_lib, d_lib = tref.fpn_rpn(m, ps)
    assert torch.equal(s_own, s_own2) and torch.equal(d_own, d_own2)        # the hand-written kernels are deterministic
    scale = max(1.0, float(s_lib.abs().max().item()), float(d_lib.abs().max().item()))
    assert float((s_own - s_lib).abs().max().item()) <= 1e-4 * scale
    assert float((d_own - d_lib).abs().max().item()) <= 1e-4 * scale


@pytest.mark.gpu
@pytest.mark.parametrize('B,A,cin,cout,shapes', [(2, 3, 256, 512, [(50, 84), (25, 42), (13, 21), (7, 11), (4, 6)]),
                                               (1, 3, 64, 512, [(100, 167)]), (3, 4, 128, 256, [(20, 31), (9, 9)]),
                                               (1, 1, 64, 512, [(37, 53)])])
def test_rpn_head_fused_matches_the_two_pass_form(B, A, cin, cout, shapes):
    """odet_rpn_head_fused_f16 (the RpnHead's 3x3 convolution with the two 1x1 convolutions in its epilogue, all levels
    in one launch, the 512-channel activation never written): EXACT on integer-valued data chosen so that every
    intermediate is an integer below 2048 (float16-exact: the fused form rounds relu(conv + bias) once, the two-pass form
    rounds the convolution first); on random data within float16 rounding of the float32 torch formulation and of the
    two-pass form (ops.conv3x3_f16_levels + ops.rpn_head_tail); deterministic, and every element of the pre-filled
    output arrays is overwritten (the channel tiles' partial sums meet in a second small launch, no atomics)."""
    from tf_eager_object_detection_amd import ops
    g = torch.Generator(device='cuda')
    g.manual_seed(B * 100 + A)
    n = sum(h * w for h, w in shapes) * A

    def reference(xs, w3, b3, w1, b1):
        sc, dl = [], []
        for x in xs:
            t = F.relu(F.conv2d(x.permute(0, 3, 1, 2).float(), w3.float(), b3.float(), 1, 1)).half().float()
            o = F.conv2d(t, w1.float().reshape(6 * A, cout, 1, 1), b1.float()).permute(0, 2, 3, 1)      # [B,h,w,6A]
            sc.append(o[..., :2 * A].reshape(B, -1, 2))
            dl.append(o[..., 2 * A:].reshape(B, -1, 4))
        return torch.cat(sc, 1), torch.cat(dl, 1)

    # integer data: |conv| <= 9 * cin * 1 * 1 would exceed 2048 for cin = 256 -> sparse operands keep it small
    xs = [(torch.randint(0, 100, (B, h, w, cin), device='cuda', generator=g) < 3).half() for h, w in shapes]
    w3 = (torch.randint(0, 100, (cout, cin, 3, 3), device='cuda', generator=g) < 4).half()
    w3 = (w3 * torch.randint(-2, 3, w3.shape, device='cuda', generator=g).half()).contiguous(memory_format=torch.channels_last)
    b3 = torch.randint(-3, 4, (cout,), device='cuda', generator=g).half()
    w1 = ((torch.randint(0, 100, (6 * A, cout), device='cuda', generator=g) < 10).half()
          * torch.randint(-2, 3, (6 * A, cout), device='cuda', generator=g).half())
    b1 = torch.randint(-3, 4, (6 * A,), device='cuda', generator=g).half()
    scores = torch.full((B, n, 2), 7.0, device='cuda')
    deltas = torch.full((B, n, 4), 7.0, device='cuda')
    ops.rpn_head_fused(xs, w3, b3, w1, b1, A, scores, deltas)
    ws, wd = reference(xs, w3, b3, w1, b1)
    assert float(ws.abs().max().item()) < 2048 and torch.equal(scores, ws) and torch.equal(deltas, wd)
    # random data
    xs = [(torch.randn((B, h, w, cin), device='cuda', generator=g) * 0.5).half() for h, w in shapes]
    w3 = (torch.randn((cout, cin, 3, 3), device='cuda', generator=g) * 0.02).half().contiguous(memory_format=torch.channels_last)
    b3 = (torch.randn(cout, device='cuda', generator=g) * 0.1).half()
    w1 = (torch.randn((6 * A, cout), device='cuda', generator=g) * 0.05).half()
    b1 = torch.randn(6 * A, device='cuda', generator=g).half()
    ops.rpn_head_fused(xs, w3, b3, w1, b1, A, scores, deltas)
    ws, wd = reference(xs, w3, b3, w1, b1)
    torch.testing.assert_close(scores, ws, rtol=3e-3, atol=3e-3)
    torch.testing.assert_close(deltas, wd, rtol=3e-3, atol=3e-3)
    s2, d2 = torch.empty_like(scores), torch.empty_like(deltas)
    ops.rpn_head_fused(xs, w3, b3, w1, b1, A, s2, d2)
    assert torch.equal(s2, scores) and torch.equal(d2, deltas)
    if cout == 512 and A <= 4:      # the two-pass form of the product (ops.rpn_head_tail takes 512 channels)
        convs = ops.conv3x3_f16_levels(xs, w3)
        s3, d3 = torch.empty_like(scores), torch.empty_like(deltas)
        off = 0
        for (h, w), c in zip(shapes, convs):
            ops.rpn_head_tail(c, b3, w1, b1, A, s3, d3, off)
            off += h * w * A
        torch.testing.assert_close(scores, s3, rtol=3e-3, atol=3e-3)
        torch.testing.assert_close(deltas, d3, rtol=3e-3, atol=3e-3)


@pytest.mark.gpu
@pytest.mark.parametrize('B,H,W,cin,n3,res,cmid', [(1, 13, 21, 256, 1024, True, 256), (2, 25, 42, 256, 1024, True, 256),
                                                   (1, 50, 84, 64, 256, False, 256), (3, 7, 5, 128, 64, True, 256),
                                                   (1, 57, 100, 256, 512, True, 256), (2, 25, 42, 128, 512, True, 128),
                                                   (1, 33, 47, 64, 256, True, 64), (1, 9, 70, 128, 192, False, 128),
                                                   (3, 6, 5, 64, 64, True, 64)])
def test_conv3x3_conv1x1_fused_block_tail(B, H, W, cin, n3, res, cmid):
    """odet_conv3x3_conv1x1_f16 (a bottleneck's 3x3 convolution with the block's last 1x1 convolution, bias, shortcut and
    ReLU in its epilogue; the 64 / 128 / 256-channel activation between them lives in LDS only): EXACT on integer-valued data whose
    intermediates stay below 2048, within float16 rounding of the float32 formulation on random data, and of the
    two-launch form of the product (ops.conv3x3_f16 + ops.conv1x1_f16(in_bias=...))."""
    from tf_eager_object_detection_amd import ops
    g = torch.Generator(device='cuda')
    g.manual_seed(B * 100 + H)

    def reference(x, w2, b2, w3, b3, r):
        t = F.relu(F.conv2d(x.permute(0, 3, 1, 2).float(), w2.float(), b2.float(), 1, 1)).half().float()
        o = F.conv2d(t, w3.float().reshape(n3, cmid, 1, 1), b3.float()).permute(0, 2, 3, 1)
        if r is not None:
            o = o + r.float()
        return F.relu(o)

    x = (torch.randint(0, 100, (B, H, W, cin), device='cuda', generator=g) < 3).half()
    w2 = (torch.randint(0, 100, (cmid, cin, 3, 3), device='cuda', generator=g) < 4).half()
    w2 = (w2 * torch.randint(-2, 3, w2.shape, device='cuda', generator=g).half()).contiguous(memory_format=torch.channels_last)
    b2 = torch.randint(-3, 4, (cmid,), device='cuda', generator=g).half()
    w3 = ((torch.randint(0, 100, (n3, cmid), device='cuda', generator=g) < 10).half()
          * torch.randint(-2, 3, (n3, cmid), device='cuda', generator=g).half())
    b3 = torch.randint(-3, 4, (n3,), device='cuda', generator=g).half()
    r = torch.randint(-4, 5, (B, H, W, n3), device='cuda', generator=g).half() if res else None
    got = ops.conv3x3_conv1x1_f16(x, w2, b2, w3, b3, residual=r, relu=True)
    want = reference(x, w2, b2, w3, b3, r)
    assert float(want.abs().max().item()) < 2048 and torch.equal(got.float(), want)
    x = (torch.randn((B, H, W, cin), device='cuda', generator=g) * 0.5).half()
    w2 = (torch.randn((cmid, cin, 3, 3), device='cuda', generator=g) * 0.02).half().contiguous(memory_format=torch.channels_last)
    b2 = (torch.randn(cmid, device='cuda', generator=g) * 0.1).half()
    w3 = (torch.randn((n3, cmid), device='cuda', generator=g) * 0.05).half()
    b3 = torch.randn(n3, device='cuda', generator=g).half()
    r = torch.randn((B, H, W, n3), device='cuda', generator=g).half() if res else None
    got = ops.conv3x3_conv1x1_f16(x, w2, b2, w3, b3, residual=r, relu=True)
    torch.testing.assert_close(got.float(), reference(x, w2, b2, w3, b3, r), rtol=4e-3, atol=4e-3)
    out = torch.empty_like(got)
    ops.conv3x3_conv1x1_f16(x, w2, b2, w3, b3, residual=r, relu=True, out=out)
    assert torch.equal(out, got)
    two = ops.conv1x1_f16(ops.conv3x3_f16(x, w2), w3, b3, residual=r, relu=True, in_bias=b2)
    torch.testing.assert_close(got.float(), two.float(), rtol=4e-3, atol=4e-3)


@pytest.mark.gpu
@pytest.mark.parametrize('B,H,W,dt', [(1, 64, 96, torch.float32), (2, 61, 75, torch.float16), (1, 600, 800, torch.float32),
                                      (3, 1, 1, torch.float32), (2, 9, 33, torch.float16), (1, 8, 32, torch.float32)])
def test_conv3x3_rgb_first_convolution(B, H, W, dt):
    """odet_conv3x3_rgb_f16 (VGG16's first convolution, vgg16_faster_rcnn.py:260-342: Conv2D(64, 3x3, 'same') + ReLU straight
    from the 3-channel image): EXACT on integer-valued data, within float16 rounding of the float32 torch convolution on
    random data; tiles that hang over the right / bottom edge, a one-pixel image, without the ReLU."""
    from tf_eager_object_detection_amd import ops
    g = torch.Generator(device='cuda')
    g.manual_seed(H * 11 + W)

    def reference(img, w, b, relu):
        y = F.conv2d(img.permute(0, 3, 1, 2).half().float(), w.float(), b.float(), 1, 1)
        return (F.relu(y) if relu else y).permute(0, 2, 3, 1)

    img = torch.randint(-3, 4, (B, H, W, 3), device='cuda', generator=g).to(dt)
    w = torch.randint(-2, 3, (64, 3, 3, 3), device='cuda', generator=g).half()
    b = torch.randint(-3, 4, (64,), device='cuda', generator=g).half()
    pw = ops.conv3x3_rgb_pack_weights(w)
    for relu in (True, False):
        got = ops.conv3x3_rgb(img, pw, b, relu=relu)
        want = reference(img, w, b, relu)
        assert got.shape == want.shape and torch.equal(got.float(), want)
    assert torch.equal(ops.conv3x3_rgb_pack_weights(w.contiguous(memory_format=torch.channels_last)), pw)
    img = (torch.randn((B, H, W, 3), device='cuda', generator=g) * 50).to(dt)
    w = (torch.randn((64, 3, 3, 3), device='cuda', generator=g) * 0.05).half()
    b = torch.randn(64, device='cuda', generator=g).half()
    got = ops.conv3x3_rgb(img, ops.conv3x3_rgb_pack_weights(w), b)
    torch.testing.assert_close(got.float(), reference(img, w, b, True), rtol=4e-3, atol=2e-2)


@pytest.mark.gpu
@pytest.mark.parametrize('B,H,W,dt', [(1, 64, 96, torch.float32), (2, 61, 75, torch.float16), (1, 800, 1333, torch.float32),
                                      (3, 7, 9, torch.float32), (1, 33, 17, torch.float16)])
def test_stem_conv7_pool3_fused(B, H, W, dt):
    """odet_stem_conv7_pool3_f16 (the ResNet stem in one launch: pad 3 + 7x7/2 convolution + folded-BN bias + ReLU + pad 1
    + 3x3/2 max-pooling, convolution on the matrix cores, nothing but the image read and the pooled map written): EXACT
    on integer-valued data (products and sums are small integers), within float16 rounding of the float32 torch
    formulation on random data; odd sizes (tiles that hang over the right / bottom edge, one-tile images)."""
    from tf_eager_object_detection_amd import ops
    g = torch.Generator(device='cuda')
    g.manual_seed(H * 7 + W)

    def reference(img, w, b):
        x = F.pad(img.permute(0, 3, 1, 2).half().float(), (3, 3, 3, 3))
        y = F.relu(F.conv2d(x, w.float(), b.float(), 2, 0)).half().float()
        return F.max_pool2d(F.pad(y, (1, 1, 1, 1)), 3, 2).permute(0, 2, 3, 1)

    img = torch.randint(-3, 4, (B, H, W, 3), device='cuda', generator=g).to(dt)
    w = torch.randint(-2, 3, (64, 3, 7, 7), device='cuda', generator=g).half()
    b = torch.randint(-3, 4, (64,), device='cuda', generator=g).half()
    pw = ops.stem_pack_weights(w)
    got = ops.stem_conv7_pool3(img, pw, b)
    want = reference(img, w, b)
    assert got.shape == want.shape and torch.equal(got.float(), want)
    # a channels_last weight tensor packs to the same thing
    assert torch.equal(ops.stem_pack_weights(w.contiguous(memory_format=torch.channels_last)), pw)
    img = (torch.randn((B, H, W, 3), device='cuda', generator=g) * 50).to(dt)
    w = (torch.randn((64, 3, 7, 7), device='cuda', generator=g) * 0.02).half()
    b = torch.randn(64, device='cuda', generator=g).half()
    got = ops.stem_conv7_pool3(img, ops.stem_pack_weights(w), b)
    torch.testing.assert_close(got.float(), reference(img, w, b), rtol=4e-3, atol=2e-2)



# ---- accuracy gate of the float16 throughput mode (evaluation/precision_gate.py) ------------------------------------------

def test_paired_map_delta_matcher_equals_the_reference_routine_cpu():
    """the per-image matcher the bootstrap re-uses gives the same mAP as evaluate_detections (the restated voc_eval,
    pinned by the reference's own routine in tests/test_evaluation.py) on imperfect detections"""
    from tf_eager_object_detection_amd.evaluation import precision_gate as pg, pascal_eval as pe
    from tf_eager_object_detection_amd import synthetic as syn
    from oracle import oracle_np as on
    rng = np.random.default_rng(5)
    dets, dets2, gb, gl = [], [], [], []
    for i in range(10):
        im = syn.eval_image(rng, num_rois=100)
        kw = dict(score_threshold=0.05, iou_threshold=0.3, max_objects_per_class=50, max_objects_per_image=50, min_size=10)
        sc = im['scores'] * rng.uniform(0.3, 1.0, im['scores'].shape).astype(np.float32)       # degrade: misses, re-ranking
        dets.append(on.eval_detect_image(sc, im['deltas'] + rng.normal(0, 1.5, im['deltas'].shape).astype(np.float32),
                                         im['rois'], im['img_scale'], im['raw_h'], im['raw_w'], **kw))
        dets2.append(on.eval_detect_image(im['scores'], im['deltas'], im['rois'], im['img_scale'], im['raw_h'], im['raw_w'], **kw))
        gb.append(im['gt_boxes'])
        gl.append(im['gt_labels'])
    present = sorted(set(int(l) for g in gl for l in g))
    for d in (dets, dets2):
        aps = pe.evaluate_detections(d, gb, gl)[1]
        want = float(np.mean([aps[j - 1] for j in present]))
        got = pg._map_from_matches(pg._image_matches(d, gb, gl, 21), np.arange(10))
        assert abs(got - want) < 1e-12
    pair = pg.paired_map_delta(dets, dets2, gb, gl, resamples=50)
    assert 0.0 < pair['map_a'] < pair['map_b'] <= 1.0 + 1e-12 and pair['delta'] > 0
    assert pair['delta_ci95'][0] <= pair['delta_boot_mean'] <= pair['delta_ci95'][1]
    # the bootstrap's re-weighting form (every detection of image i counts counts[i] times) against the expanded index list
    m, f = pg._image_matches(dets, gb, gl, 21), pg._flat_matches(pg._image_matches(dets, gb, gl, 21))
    for _ in range(20):
        idx = rng.integers(0, 10, 10)
        for m07 in (True, False):
            assert abs(pg._map_from_matches(m, idx, m07) - pg._map_weighted(f, np.bincount(idx, minlength=10), m07)) < 1e-12


def _report_gate(rec):
    print('\n  %s fp16 vs fp32 on %d scenes: mAP %.4f -> %.4f, delta %+.4f (bootstrap std %.4f, CI95 [%+.4f, %+.4f]); reproduction: '
          '%.1f %% of the fp32 detections matched, RPN kept-index agreement %.3f, median |dscore| %.1e'
          % (rec['model'], rec['images'], rec['map_fp32'], rec['map_fp16'], rec['map_delta'], rec['map_delta_bootstrap_std'],
             rec['map_delta_ci95_paired_bootstrap'][0], rec['map_delta_ci95_paired_bootstrap'][1],
             100 * rec['reproduction']['matched_fraction'], rec['rpn_kept_index_agreement_mean'], rec['median_abs_dscore']))


@pytest.mark.gpu
def test_fp16_detector_map_delta_vs_fp32_on_identical_weights_and_images():
    """BASELINE metric's second half for the mode that meets the throughput target: ResNet-101-FPN @ 800x1333, float16
    against float32 on the same seeded weights (last linear layers fitted on annotated scenes: a genuine detector,
    mAP ~0.5) and the same 1024 held-out scenes, through im_detect -> detect_image -> VOC07 mAP against the annotations.
    Bar (north star): the POINT ESTIMATE |mAP(fp16) - mAP(fp32)| <= 0.002, and the paired-bootstrap 95 % interval inside
    +-0.004 (bench.py runs 4096 scenes, where the interval's half-width is ~0.0014: profiles/r04_bench_driver_cmd.json)."""
    from tf_eager_object_detection_amd.evaluation import precision_gate as pg
    rec = pg.fp16_vs_fp32(num_images=1024, resamples=300, batch32=16, batch16=32)
    _report_gate(rec)
    assert rec['classes_scored'] == 20 and rec['map_fp32'] >= 0.3            # the fitted detector has real signal
    assert abs(rec['map_delta']) <= 0.002, rec['map_delta']
    lo, hi = rec['map_delta_ci95_paired_bootstrap']
    assert -0.004 <= lo <= hi <= 0.004, (lo, hi)
    assert rec['reproduction']['matched_fraction'] >= 0.95 and rec['rpn_kept_index_agreement_mean'] >= 0.96
    assert rec['median_abs_dscore'] <= 2e-3 and rec['median_abs_dbox_px'] <= 0.5


@pytest.mark.gpu
@pytest.mark.parametrize('family', ['c4', 'vgg16'])
def test_fp16_single_level_detectors_map_delta_vs_fp32(family):
    """the same gate for the float16 ResNet-50 C4 (800x1333) and VGG16 (600x800) detectors -- BASELINE configs 2 and 1: their
    throughput records carry accuracy evidence too.  512 held-out scenes: |delta| <= 0.002 + 2 sigma of the paired bootstrap."""
    from tf_eager_object_detection_amd.evaluation import precision_gate as pg
    rec = pg.fp16_vs_fp32(num_images=512, resamples=300, batch32=16, batch16=32, family=family)
    _report_gate(rec)
    assert rec['classes_scored'] == 20 and rec['map_fp32'] >= 0.2
    assert abs(rec['map_delta']) <= 0.002 + 2.0 * rec['map_delta_bootstrap_std'], rec['map_delta']
    assert rec['reproduction']['matched_fraction'] >= 0.93


# ---- pointwise form of the implicit-GEMM kernel (odet_pointwise_f16 / odet_lateral_merge_f16) -----------------------------

def _int_operands(g, M, K, N):
    """small-integer float16 operands whose contraction stays exactly representable: sparse x for long K"""
    ri = lambda lo, hi, *sh: torch.randint(lo, hi, sh, device='cuda', generator=g).to(torch.float16)
    x, w = ri(-2, 3, M, K), ri(-1, 2, N, K)
    if K > 512:
        keep = torch.rand((M, K), device='cuda', generator=g) < 256.0 / K
        x = x * keep.to(torch.float16)
    w[:, 0] += torch.arange(N, device='cuda').remainder(5).to(torch.float16)
    x[:, 1] += torch.arange(M, device='cuda').remainder(7).to(torch.float16)
    return x, w


@pytest.mark.gpu
@pytest.mark.parametrize('B,H,W,K,N,stride', [(2, 25, 42, 1024, 256, 1), (1, 50, 84, 512, 128, 1), (2, 50, 84, 256, 512, 2),
                                             (1, 33, 47, 512, 1024, 2), (3, 13, 21, 2048, 512, 1), (1, 1, 1000, 12544, 1024, 1),
                                             (1, 100, 167, 256, 128, 2), (2, 31, 17, 128, 64, 1), (1, 9, 11, 1024, 2048, 2)])
def test_pointwise_f16_exact_on_integer_data_and_close_on_random(B, H, W, K, N, stride):
    """odet_pointwise_f16: the bottlenecks' first / strided shortcut 1x1 convolutions, the neck's P5 convolution and the
    RoI head's dense layers (resnet_fpn.py:154-205, 292-336, 339-384) on the LDS-staged GEMM.  EXACT on small-integer
    data (any fragment / permutation / stride slip shows), within float16 rounding of a float32 torch formulation on
    random data; every epilogue combination; maps whose pixel count does not fill a tile."""
    from tf_eager_object_detection_amd import ops
    g = torch.Generator(device='cuda'); g.manual_seed(B * 7 + H + K + N + stride)
    Ho, Wo = (H + stride - 1) // stride, (W + stride - 1) // stride
    x2, w = _int_operands(g, B * H * W, K, N)
    x = x2.view(B, H, W, K)
    b = torch.randint(-8, 9, (N,), device='cuda', generator=g).to(torch.float16)
    r = torch.randint(-16, 17, (B, Ho, Wo, N), device='cuda', generator=g).to(torch.float16)
    xs = x[:, ::stride, ::stride].float()
    for res, relu, bias in ((r, True, b), (None, True, b), (r, False, b), (None, False, None)):
        want = xs @ w.float().t()
        if bias is not None:
            want = want + bias.float()
        if res is not None:
            want = want + res.float()
        if relu:
            want = torch.relu(want)
        assert float(want.abs().max()) < 2048
        got = ops.pointwise_f16(x, w.view(N, K, 1, 1), bias, res, relu, stride)
        assert got.shape == (B, Ho, Wo, N) and torch.equal(got.float(), want)
    x = torch.randn((B, H, W, K), device='cuda', generator=g).to(torch.float16)
    w = (torch.randn((N, K), device='cuda', generator=g) / K ** 0.5).to(torch.float16)
    want = torch.relu(x[:, ::stride, ::stride].float() @ w.float().t() + b.float() + r.float())
    got = ops.pointwise_f16(x, w, b, r, True, stride)
    torch.testing.assert_close(got.float(), want, rtol=2e-3, atol=4e-3)
    if H == 1 and B == 1 and stride == 1:                               # the dense-layer view
        assert torch.equal(ops.dense_f16(x.view(W, K), w, b, True), ops.pointwise_f16(x, w, b, None, True).view(W, N))
    with pytest.raises(Exception):
        ops.pointwise_f16(x[..., :64].contiguous(), w[:, :64].contiguous(), b)       # K = 64: one K-step, not served


@pytest.mark.gpu
@pytest.mark.parametrize('B,H,W,h,w_,K', [(2, 50, 84, 25, 42, 1024), (1, 100, 167, 50, 84, 512), (1, 37, 53, 19, 27, 256),
                                          (2, 20, 20, 10, 10, 128)])
def test_lateral_merge_f16_equals_lateral_then_merge(B, H, W, h, w_, K):
    """odet_lateral_merge_f16 (resnet_fpn.py:385-398 fused into the lateral convolution's epilogue): identical to
    odet_fpn_topdown_merge applied to the un-rounded lateral convolution on integer data (where the lateral map is exact
    in float16), within float16 rounding of the float32 formulation on random data."""
    from tf_eager_object_detection_amd import ops
    from tf_eager_object_detection_amd.model.fpn_detector import tf_legacy_resize_bilinear
    g = torch.Generator(device='cuda'); g.manual_seed(H * W + K)
    N = 256
    x2, w = _int_operands(g, B * H * W, K, N)
    x = x2.view(B, H, W, K)
    b = torch.randint(-8, 9, (N,), device='cuda', generator=g).to(torch.float16)
    top = torch.randint(-64, 65, (B, h, w_, N), device='cuda', generator=g).to(torch.float16)
    lat = (x.float() @ w.float().t() + b.float())
    assert float(lat.abs().max()) < 2048
    want = ops.fpn_topdown_merge(top, lat.to(torch.float16))            # the two-launch form on the exact lateral map
    got = ops.lateral_merge_f16(x, w, b, top)
    assert torch.equal(got, want)
    x = torch.randn((B, H, W, K), device='cuda', generator=g).to(torch.float16)
    w = (torch.randn((N, K), device='cuda', generator=g) / K ** 0.5).to(torch.float16)
    top = torch.randn((B, h, w_, N), device='cuda', generator=g).to(torch.float16)
    lat = x.float() @ w.float().t() + b.float()
    up = tf_legacy_resize_bilinear(top.float().permute(0, 3, 1, 2), (H, W)).permute(0, 2, 3, 1)
    want = up * 0.5 + lat * 0.5
    torch.testing.assert_close(ops.lateral_merge_f16(x, w, b, top).float(), want, rtol=2e-3, atol=4e-3)


@pytest.mark.gpu
def test_dense_f16_out_f32_last_layer():
    """odet_dense_f16_out_f32: the RoI head's score / bbox layer with float32 results -- exact on integer data, equal to
    the float32 contraction of the float16 operands on random data up to accumulation order"""
    from tf_eager_object_detection_amd import ops
    g = torch.Generator(device='cuda'); g.manual_seed(11)
    M, K, N = 1037, 1024, 128
    x, w = _int_operands(g, M, K, N)
    w[105:] = 0
    b = torch.randint(-50, 51, (N,), device='cuda', generator=g).float() * 0.25
    got = ops.dense_f16_out_f32(x, w, b)
    assert got.dtype == torch.float32 and torch.equal(got, x.float() @ w.float().t() + b)
    x = torch.randn((M, K), device='cuda', generator=g).to(torch.float16)
    w = (torch.randn((N, K), device='cuda', generator=g) / K ** 0.5).to(torch.float16)
    want = x.double() @ w.double().t() + b.double()
    torch.testing.assert_close(ops.dense_f16_out_f32(x, w, b).double(), want, rtol=1e-5, atol=1e-5)
    assert torch.equal(ops.dense_f16_out_f32(x, w, b, relu=True), torch.relu(ops.dense_f16_out_f32(x, w, b)))


@pytest.mark.gpu
@pytest.mark.parametrize('B,H,W,K1,K2,N,stride', [(2, 20, 33, 64, 64, 256, 1), (1, 50, 84, 128, 256, 512, 2),
                                                  (2, 25, 41, 256, 512, 1024, 2), (1, 13, 21, 512, 1024, 2048, 2)])
def test_pointwise_dual_f16_last_conv_and_shortcut_in_one_contraction(B, H, W, K1, K2, N, stride):
    """odet_pointwise_dual_f16 (a stage's first bottleneck, resnet_fpn.py:154-205): relu(w3 . y2 + b3 + w_sc . x(::s) + b_sc)
    as ONE contraction over the concatenated K.  EXACT on integer data, within float16 rounding of float32 torch."""
    from tf_eager_object_detection_amd import ops
    g = torch.Generator(device='cuda'); g.manual_seed(K1 + K2 + N)
    Ho, Wo = (H + stride - 1) // stride, (W + stride - 1) // stride
    a, wa = _int_operands(g, B * Ho * Wo, K1, N)
    c, wc = _int_operands(g, B * H * W, K2, N)
    if K1 + K2 > 512:
        keep = torch.rand((B * H * W, K2), device='cuda', generator=g) < 128.0 / K2
        c = c * keep.to(torch.float16)
        a = a * (torch.rand((B * Ho * Wo, K1), device='cuda', generator=g) < 128.0 / K1).to(torch.float16)
    a, c = a.view(B, Ho, Wo, K1), c.view(B, H, W, K2)
    b = torch.randint(-8, 9, (N,), device='cuda', generator=g).to(torch.float16)
    w = torch.cat([wa, wc], 1).contiguous()
    want = a.float() @ wa.float().t() + c[:, ::stride, ::stride].float() @ wc.float().t() + b.float()
    assert float(want.abs().max()) < 2048
    assert torch.equal(ops.pointwise_dual_f16(a, c, w, b, stride, relu=True).float(), torch.relu(want))
    assert torch.equal(ops.pointwise_dual_f16(a, c, w, None, stride, relu=False).float(), want - b.float())
    a = torch.randn((B, Ho, Wo, K1), device='cuda', generator=g).to(torch.float16)
    c = torch.randn((B, H, W, K2), device='cuda', generator=g).to(torch.float16)
    w = (torch.randn((N, K1 + K2), device='cuda', generator=g) / (K1 + K2) ** 0.5).to(torch.float16)
    want = torch.relu(a.float() @ w[:, :K1].float().t() + c[:, ::stride, ::stride].float() @ w[:, K1:].float().t() + b.float())
    torch.testing.assert_close(ops.pointwise_dual_f16(a, c, w, b, stride).float(), want, rtol=2e-3, atol=4e-3)


# ---- float32 forms (csrc/conv_f32.hip): the parity mode's dense layers on exact-float32 matrix instructions --------------

@pytest.mark.gpu
@pytest.mark.parametrize('B,H,W,K,N,stride', [(2, 25, 42, 1024, 256, 1), (1, 50, 84, 256, 512, 2), (1, 33, 47, 64, 64, 1),
                                             (2, 13, 21, 2048, 512, 1), (1, 1, 700, 12544, 1024, 1), (1, 100, 167, 256, 128, 2),
                                             (1, 40, 61, 160, 64, 1)])
def test_pointwise_f32_exact_on_integer_data_and_equal_to_float64_on_random(B, H, W, K, N, stride):
    """odet_pointwise_f32: EXACT on integer data (float32 holds every partial sum), within float32 accumulation error of a
    float64 contraction on random data; shortcut and ReLU epilogues; strided rows."""
    from tf_eager_object_detection_amd import ops
    g = torch.Generator(device='cuda'); g.manual_seed(B + H + K + N + stride)
    Ho, Wo = (H + stride - 1) // stride, (W + stride - 1) // stride
    ri = lambda lo, hi, *sh: torch.randint(lo, hi, sh, device='cuda', generator=g).float()
    x, w, b, r = ri(-8, 9, B, H, W, K), ri(-4, 5, N, K), ri(-100, 101, N), ri(-1000, 1001, B, Ho, Wo, N)
    xs = x[:, ::stride, ::stride]
    for res, relu, bias in ((r, True, b), (None, False, b), (r, False, None)):
        want = xs.double() @ w.double().t()
        if bias is not None:
            want = want + bias.double()
        if res is not None:
            want = want + res.double()
        if relu:
            want = torch.relu(want)
        assert float(want.abs().max()) < 2 ** 24
        got = ops.pointwise(x, w, bias, res, relu, stride)
        assert got.dtype == torch.float32 and torch.equal(got.double(), want)
    x = torch.randn((B, H, W, K), device='cuda', generator=g)
    w = torch.randn((N, K), device='cuda', generator=g) / K ** 0.5
    want = torch.relu(x[:, ::stride, ::stride].double() @ w.double().t() + b.double() + r.double())
    got = ops.pointwise(x, w, b, r, True, stride).double()
    assert float((got - want).abs().max()) <= 2e-5 * float(want.abs().max())


@pytest.mark.gpu
def test_float32_lateral_merge_dual_and_small_tile_convolutions():
    """the other float32 forms: lateral convolution + top-down merge in one launch is BIT-IDENTICAL to the merge launch on
    the convolution's result; a stage's first bottleneck (last convolution + shortcut as one contraction) and the 64 /
    128-channel 3x3 tiles exact on integer data; the stem's patch matrix + GEMM equals the library convolution."""
    from tf_eager_object_detection_amd import ops
    g = torch.Generator(device='cuda'); g.manual_seed(3)
    x = torch.randn((2, 37, 53, 512), device='cuda', generator=g)
    w = torch.randn((256, 512), device='cuda', generator=g) / 512 ** 0.5
    b = torch.randn(256, device='cuda', generator=g)
    top = torch.randn((2, 19, 27, 256), device='cuda', generator=g)
    assert torch.equal(ops.lateral_merge(x, w, b, top), ops.fpn_topdown_merge(top, ops.pointwise(x, w, b)))
    ri = lambda lo, hi, *sh: torch.randint(lo, hi, sh, device='cuda', generator=g).float()
    a, c = ri(-8, 9, 1, 25, 42, 128), ri(-8, 9, 1, 50, 84, 256)
    wa, wc, bb = ri(-4, 5, 512, 128), ri(-4, 5, 512, 256), ri(-50, 51, 512)
    want = torch.relu(a.double() @ wa.double().t() + c[:, ::2, ::2].double() @ wc.double().t() + bb.double())
    assert torch.equal(ops.pointwise_dual(a, c, torch.cat([wa, wc], 1).contiguous(), bb, 2, relu=True).double(), want)
    for cin, cout in ((64, 64), (128, 128), (64, 192)):
        xi, wi, bi = ri(-4, 5, 2, 19, 23, cin), ri(-3, 4, cout, cin, 3, 3), ri(-20, 21, cout)
        want = torch.relu(F.conv2d(xi.permute(0, 3, 1, 2).double(), wi.double(), bi.double(), 1, 1)).permute(0, 2, 3, 1)
        got = ops.conv3x3_f32(xi, wi.contiguous(memory_format=torch.channels_last), bi, relu=True)
        assert torch.equal(got.double(), want)
    img = torch.randn((2, 61, 83, 3), device='cuda', generator=g)
    w7 = torch.randn((64, 3, 7, 7), device='cuda', generator=g) * 0.1
    w160 = torch.zeros((64, 160), device='cuda')
    w160[:, :147] = w7.permute(0, 2, 3, 1).reshape(64, 147)
    got = ops.pointwise(ops.stem_patches_f32(img), w160, None)
    want = F.conv2d(img.permute(0, 3, 1, 2).double(), w7.double(), None, 2, 3).permute(0, 2, 3, 1)
    assert got.shape == want.shape and float((got.double() - want).abs().max()) < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize('family', ['fpn', 'c4', 'vgg16'])
def test_float32_detector_pass_runs_no_library_convolution_or_gemm(monkeypatch, family):
    """the parity mode's dense path is this repository's kernels end to end in all three model families: a float32 pass calls
    neither torch's convolution nor its GEMMs (the routes would fall back to them silently otherwise)"""
    import torch.nn.functional as Fn
    from tf_eager_object_detection_amd.model.fpn_detector import ResNetFpnDetector
    from tf_eager_object_detection_amd.model.frcnn_detector import ResNetC4Detector, Vgg16Detector
    torch.manual_seed(1)
    shape = (320, 480)
    if family == 'fpn':
        m = ResNetFpnDetector(50, 21, shape, 300, dtype=torch.float32, max_batch=2).prepare()
    elif family == 'c4':
        m = ResNetC4Detector(50, 21, shape, 64, dtype=torch.float32, max_batch=2).prepare()
    else:
        m = Vgg16Detector(21, shape, 64, dtype=torch.float32, max_batch=2).prepare()
    img = torch.randn((2,) + shape + (3,), device='cuda') * 50
    out_ref = m(img)
    calls = _count_library_calls(monkeypatch)
    out = m(img)
    assert calls == [], calls
    assert int(out[0][3].item()) == int(out_ref[0][3].item())


@pytest.mark.gpu
def test_rgb_patches3x3_f32_first_convolution():
    """odet_rgb_patches3x3_f32 + the exact-float32 pointwise GEMM = VGG16's first convolution in the parity mode: exact on
    integer data, within 2e-5 of a float64 convolution on random data (odd sizes, batch 2)"""
    from tf_eager_object_detection_amd import ops
    g = torch.Generator(device='cuda'); g.manual_seed(5)
    for B, H, W in ((2, 17, 23), (1, 64, 96)):
        img = torch.randint(-3, 4, (B, H, W, 3), device='cuda', generator=g).float()
        w = torch.randint(-2, 3, (64, 3, 3, 3), device='cuda', generator=g).float()
        b = torch.randint(-3, 4, (64,), device='cuda', generator=g).float()
        w64 = torch.zeros((64, 64), device='cuda'); w64[:, :27] = w.permute(0, 2, 3, 1).reshape(64, 27)
        got = ops.pointwise(ops.rgb_patches3x3_f32(img), w64, b, None, True)
        want = F.relu(F.conv2d(img.permute(0, 3, 1, 2).double(), w.double(), b.double(), 1, 1)).permute(0, 2, 3, 1)
        assert torch.equal(got.double(), want)
        img = torch.randn((B, H, W, 3), device='cuda', generator=g) * 50
        w = torch.randn((64, 3, 3, 3), device='cuda', generator=g) * 0.05
        w64 = torch.zeros((64, 64), device='cuda'); w64[:, :27] = w.permute(0, 2, 3, 1).reshape(64, 27)
        got = ops.pointwise(ops.rgb_patches3x3_f32(img), w64, b, None, True)
        want = F.relu(F.conv2d(img.permute(0, 3, 1, 2).double(), w.double(), b.double(), 1, 1)).permute(0, 2, 3, 1)
        assert float((got.double() - want).abs().max()) < 2e-5 * max(1.0, float(want.abs().max()))


def _count_library_calls(monkeypatch):
    """intercepts torch's convolution / GEMM entry points; returns the list their names are appended to"""
    import torch.nn.functional as Fn
    calls = []
    for name in ('conv2d', 'linear'):
        real = getattr(Fn, name)
        monkeypatch.setattr(Fn, name, lambda *a, _r=real, _n=name, **k: calls.append(_n) or _r(*a, **k))
    for name in ('addmm', '_addmm_activation', 'matmul', 'mm', 'bmm', 'conv2d'):
        real = getattr(torch, name)
        monkeypatch.setattr(torch, name, lambda *a, _r=real, _n=name, **k: calls.append(_n) or _r(*a, **k))
    return calls


@pytest.mark.gpu
@pytest.mark.parametrize('family,shape,batch', [('fpn', (320, 480), 2), ('c4', (320, 480), 2), ('vgg16', (320, 480), 2),
                                                ('fpn', (800, 1333), 1), ('fpn', (800, 1333), 4)])
def test_float16_detector_passes_run_no_library_convolution_or_gemm(monkeypatch, family, shape, batch):
    """the float16 dense paths of all three model families (ResNet-FPN, ResNet-C4, VGG16 Faster R-CNN) are this repository's
    kernels end to end AT EVERY MAP SIZE -- small maps (every layer type), and the BASELINE configs' own shape and batch size
    (1 x 3 x 800 x 1333, ResNet-101: conv4 / conv5 / the small neck levels on the ring form of the implicit GEMM): a pass calls
    neither torch's convolution nor its GEMMs, and the detectors have no route that could (model/fpn_detector.py)."""
    from tf_eager_object_detection_amd.model.fpn_detector import ResNetFpnDetector
    from tf_eager_object_detection_amd.model.frcnn_detector import ResNetC4Detector, Vgg16Detector
    torch.manual_seed(2)
    depth = 101 if shape == (800, 1333) else 50
    if family == 'fpn':
        m = ResNetFpnDetector(depth, 21, shape, 300, dtype=torch.float16, max_batch=batch).prepare()
    elif family == 'c4':
        m = ResNetC4Detector(depth, 21, shape, 64, dtype=torch.float16, max_batch=batch).prepare()
    else:
        m = Vgg16Detector(21, shape, 64, dtype=torch.float16, max_batch=batch).prepare()
    img = torch.randn((batch,) + shape + (3,), device='cuda') * 50
    m(img)
    calls = _count_library_calls(monkeypatch)
    out = m(img)
    assert calls == [], calls
    assert all(int(f) == 1 for f in m._steps.nms_done_all[:batch].tolist())
    assert torch.isfinite(out[0][0]).all()


@pytest.mark.gpu
def test_detectors_have_no_library_or_cpu_route():
    """a layer no kernel takes raises (no silent library convolution, no CPU formulation inside the package)"""
    from tf_eager_object_detection_amd.model import fpn_detector as fd
    import inspect
    src = inspect.getsource(fd) + inspect.getsource(__import__('tf_eager_object_detection_amd.model.frcnn_detector', fromlist=['x']))
    for needle in ('F.conv2d', 'F.linear', 'torch.addmm', '_addmm_activation', 'torch.nn.functional', 'os.environ', 'getenv'):
        assert needle not in src, needle
    c = fd._conv(48, 64, 3, 1, 1).cuda().half()                          # 48 input channels: not a multiple of 64
    x = torch.zeros(1, 48, 8, 8, device='cuda', dtype=torch.float16).contiguous(memory_format=torch.channels_last)
    with pytest.raises(RuntimeError, match='no kernel'):
        fd._conv_epi(c, x)
    m = fd.ResNetFpnDetector(50, 21, (64, 64), 10, dtype=torch.float32)
    with pytest.raises(RuntimeError, match='no kernel'):
        m.features(torch.zeros(1, 64, 64, 3))                            # CPU tensors


@pytest.mark.gpu
@pytest.mark.parametrize('dt,form', [(torch.float32, 'exact'), (torch.float32, 'x3'), (torch.float32, 'x2'), (torch.float16, 'exact')],
                         ids=['float32-exact', 'float32-x3', 'float32-x2', 'float16'])
def test_detector_dense_parts_match_the_plain_torch_formulation(dt, form):
    """features / rpn / roi_head of the ResNet-FPN detector (this repository's kernels) against the plain-torch formulation of
    the same modules (library convolutions, float32 arithmetic on the same weights); float32 in both of its forms: the
    exact-float32 matrix instructions and the split-precision form (three bfloat16 limbs) under the SAME tolerance"""
    from tf_eager_object_detection_amd.model.fpn_detector import ResNetFpnDetector
    torch.manual_seed(11)
    shape = (192, 256)
    m = ResNetFpnDetector(50, 21, shape, 64, dtype=dt, max_batch=2, f32_form=form).prepare()
    img = torch.randn((2,) + shape + (3,), device='cuda') * 40
    with torch.no_grad():
        ps = m.features(img)
        sc, dl = m.rpn(ps)
        ref = ResNetFpnDetector(50, 21, shape, 64, dtype=torch.float32, max_batch=2)
        ref.load_state_dict({k: v.float() for k, v in m.state_dict().items()})
        ref = ref.cuda().eval()
        ps_ref = tref.fpn_features(ref, img)
        sc_ref, dl_ref = tref.fpn_rpn(ref, [p.float() for p in ps])       # (the head on the SAME maps: its own error only)
    tol = 2e-4 if dt == torch.float32 else 3e-2
    for a, b in zip(ps, ps_ref):
        assert a.shape == b.shape
        assert float((a.float() - b).abs().max()) <= tol * max(1.0, float(b.abs().max()))
    assert float((sc - sc_ref).abs().max()) <= tol * max(1.0, float(sc_ref.abs().max()))
    assert float((dl - dl_ref).abs().max()) <= tol * max(1.0, float(dl_ref.abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize('B,H,W,cin,cout', [(2, 37, 45, 64, 64), (1, 64, 96, 64, 128), (3, 9, 7, 128, 256), (1, 1, 1, 64, 64),
                                            (2, 150, 200, 256, 256), (1, 75, 100, 512, 512)])
def test_conv3x3_relu_pool2_fused(B, H, W, cin, cout):
    """odet_conv3x3_relu_pool2_f16 (a VGG16 stage's last convolution with its 2x2 / 2 'same' max-pooling in the launch):
    IDENTICAL to pooling the separately computed relu(conv + bias) map -- rounding commutes with the maximum -- on random
    data; odd sizes (windows that hang over the bottom / right edge), every channel-tile width"""
    from tf_eager_object_detection_amd import ops
    g = torch.Generator(device='cuda'); g.manual_seed(H * 13 + W)
    x = torch.randn((B, H, W, cin), device='cuda', generator=g).half()
    w = (torch.randn((cout, cin, 3, 3), device='cuda', generator=g) * (9 * cin) ** -0.5).half().contiguous(memory_format=torch.channels_last)
    b = torch.randn(cout, device='cuda', generator=g).half()          # (positive biases: a phantom pixel must not win)
    got = ops.conv3x3_relu_pool2_f16(x, w, b)
    full = ops.conv3x3_f16(x, w, b, relu=True)
    want = F.max_pool2d(full.permute(0, 3, 1, 2).float(), 2, 2, ceil_mode=True).permute(0, 2, 3, 1)
    assert got.shape == want.shape and torch.equal(got.float(), want)


@pytest.mark.gpu
@pytest.mark.parametrize('family', ['fpn', 'vgg16'])
def test_float32_patch_matrix_goes_through_in_groups_below_4_gib(monkeypatch, family):
    """the float32 mode's first convolution runs as a GEMM on a patch matrix addressed with 32-bit offsets: a batch whose
    patches would exceed the limit goes through in groups of images -- same features as in one piece"""
    from tf_eager_object_detection_amd.model import fpn_detector as fd
    from tf_eager_object_detection_amd.model.frcnn_detector import Vgg16Detector
    torch.manual_seed(3)
    shape = (96, 128)
    m = (fd.ResNetFpnDetector(50, 21, shape, 64, dtype=torch.float32, max_batch=3) if family == 'fpn'
         else Vgg16Detector(21, shape, 32, dtype=torch.float32, max_batch=3)).prepare()
    img = torch.randn((3,) + shape + (3,), device='cuda') * 50
    with torch.no_grad():
        whole = m.features(img)
        per_image = (48 * 64 * 160 * 4) if family == 'fpn' else (96 * 128 * 64 * 4)
        monkeypatch.setattr(fd, '_PATCH_BYTES_MAX', per_image + 1)            # one image per group
        parts = m.features(img)
    whole = whole if isinstance(whole, (list, tuple)) else [whole]
    parts = parts if isinstance(parts, (list, tuple)) else [parts]
    assert len(whole) == len(parts) and all(torch.equal(a, b) for a, b in zip(whole, parts))


def _x3_and_exact(fn, form='x3'):
    from tf_eager_object_detection_amd import ops
    with ops.f32_form('exact'):
        a = fn()
    with ops.f32_form(form):
        b = fn()
    return a, b


SPLIT_FORMS = ['x3', 'x2']          # three bfloat16 limbs / two float16 limbs (csrc/conv_x3.hip, NL = 3 / 2)


@pytest.mark.gpu
@pytest.mark.parametrize('form', SPLIT_FORMS)
@pytest.mark.parametrize('B,H,W,cin,cout', [(2, 25, 42, 256, 512), (1, 50, 84, 64, 64), (3, 13, 21, 512, 512), (1, 100, 167, 128, 128),
                                             (2, 31, 45, 32, 192), (1, 7, 5, 96, 64), (1, 200, 334, 64, 64)])
def test_conv3x3_split_precision_form(B, H, W, cin, cout, form):
    """odet_conv3x3_x3 / _x2 (csrc/conv_x3.hip: float32 operands as three bfloat16 limbs, six products per k -- or two float16
    limbs h + l * 2^-11, three products per k -- float32 accumulation):
    EXACT on integer-valued data (every limb product and every partial sum is an integer below 2^24) and, on random data, as
    close to the float64 convolution as the exact-float32 form is (within float32 rounding of the float64
    truth) -- all forms share memory layout, tiling rules and epilogue"""
    from tf_eager_object_detection_amd import ops
    g = torch.Generator(device='cuda'); g.manual_seed(H * 7 + cin)
    xi = torch.randint(-3, 4, (B, H, W, cin), device='cuda', generator=g).float()
    wi = torch.randint(-2, 3, (cout, cin, 3, 3), device='cuda', generator=g).float().contiguous(memory_format=torch.channels_last)
    bi = torch.randint(-3, 4, (cout,), device='cuda', generator=g).float()
    want = F.relu(F.conv2d(xi.permute(0, 3, 1, 2).double(), wi.double(), bi.double(), 1, 1)).permute(0, 2, 3, 1)
    ex, x3 = _x3_and_exact(lambda: ops.conv3x3_f32(xi, wi, bi, relu=True), form)
    assert torch.equal(ex.double(), want) and torch.equal(x3.double(), want)
    x = torch.randn((B, H, W, cin), device='cuda', generator=g) * 5
    w = (torch.randn((cout, cin, 3, 3), device='cuda', generator=g) * (9 * cin) ** -0.5).contiguous(memory_format=torch.channels_last)
    b = torch.randn(cout, device='cuda', generator=g)
    want = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), b.double(), 1, 1).permute(0, 2, 3, 1)
    ex, x3 = _x3_and_exact(lambda: ops.conv3x3_f32(x, w, b), form)
    rms = float(want.pow(2).mean().sqrt())
    e_ex, e_x3 = float((ex.double() - want).abs().max()) / rms, float((x3.double() - want).abs().max()) / rms
    assert e_x3 <= max(2.0 * e_ex, 2e-6) and e_x3 < 2e-5, (e_ex, e_x3)
    if form == 'x3':
        # huge and tiny magnitudes: the limbs keep float32's exponent range (bfloat16), nothing overflows or flushes
        ex, x3 = _x3_and_exact(lambda: ops.conv3x3_f32(x * 1e18, w * 1e-20, None))
        want = F.conv2d((x * 1e18).permute(0, 3, 1, 2).double(), (w * 1e-20).double(), None, 1, 1).permute(0, 2, 3, 1)
        assert float((x3.double() - want).abs().max()) <= 2e-5 * float(want.abs().max())
    else:
        # float16 limbs: the WEIGHTS' scale is free (the planes hold w * 2^w_exp), the activations must lie inside float16's
        # range -- up to 65504 at full accuracy, beyond it infinities / NaN, not a wrong finite number
        xs = x * (6.0e4 / float(x.abs().max()))
        ex, x2 = _x3_and_exact(lambda: ops.conv3x3_f32(xs, w * 1e-20, None), form)
        want = F.conv2d(xs.permute(0, 3, 1, 2).double(), (w * 1e-20).double(), None, 1, 1).permute(0, 2, 3, 1)
        assert float((x2.double() - want).abs().max()) <= 2e-6 * float(want.abs().max())
        ex, x2 = _x3_and_exact(lambda: ops.conv3x3_f32(x * 1e-3, w * 1e12, None), form)
        want = F.conv2d((x * 1e-3).permute(0, 3, 1, 2).double(), (w * 1e12).double(), None, 1, 1).permute(0, 2, 3, 1)
        assert float((x2.double() - want).abs().max()) <= 2e-6 * float(want.abs().max())
        xo = x.clone()
        xo[0, H // 2, W // 2, 3] = 7.0e4
        _, x2 = _x3_and_exact(lambda: ops.conv3x3_f32(xo, w, None), form)
        hit = x2[0, max(H // 2 - 1, 0):H // 2 + 2, max(W // 2 - 1, 0):W // 2 + 2]
        assert not torch.isfinite(hit).any() and torch.isfinite(x2).sum() == x2.numel() - hit.numel()


@pytest.mark.gpu
@pytest.mark.parametrize('form', SPLIT_FORMS)
def test_pointwise_split_precision_forms(form):
    """the 1x1 / dense / strided / two-source / lateral-merge / shortcut forms of the split-precision kernel against float64:
    exact on integers, as close as the exact-float32 form on random data; multi-level launch of the 3x3 form"""
    from tf_eager_object_detection_amd import ops
    g = torch.Generator(device='cuda'); g.manual_seed(9)

    def close(fn, want, what):
        ex, x3 = _x3_and_exact(fn, form)
        rms = max(float(want.pow(2).mean().sqrt()), 1e-30)
        e_ex, e_x3 = float((ex.double() - want).abs().max()) / rms, float((x3.double() - want).abs().max()) / rms
        assert e_x3 <= max(2.0 * e_ex, 2e-6) and e_x3 < 3e-5, (what, e_ex, e_x3)
    for B, H, W, cin, cout, stride in ((2, 25, 42, 1024, 256, 1), (1, 50, 84, 256, 512, 2), (3, 9, 7, 2048, 512, 1), (1, 33, 47, 64, 256, 1),
                                       (1, 1, 1000, 12544, 1024, 1), (2, 40, 30, 160, 64, 1)):
        x = torch.randn((B, H, W, cin), device='cuda', generator=g) * 3
        w = torch.randn((cout, cin), device='cuda', generator=g) * cin ** -0.5
        b = torch.randn(cout, device='cuda', generator=g)
        Ho, Wo = (H + stride - 1) // stride, (W + stride - 1) // stride
        res = torch.randn((B, Ho, Wo, cout), device='cuda', generator=g)
        want = torch.relu(F.conv2d(x.permute(0, 3, 1, 2).double(), w.double()[:, :, None, None], b.double(), stride).permute(0, 2, 3, 1)
                          + res.double())
        close(lambda: ops.pointwise(x, w, b, res, True, stride), want, ('pointwise', B, H, W, cin, cout, stride))
    # integers: exact
    xi = torch.randint(-3, 4, (2, 13, 17, 256), device='cuda', generator=g).float()
    wi = torch.randint(-2, 3, (128, 256), device='cuda', generator=g).float()
    want = F.conv2d(xi.permute(0, 3, 1, 2).double(), wi.double()[:, :, None, None]).permute(0, 2, 3, 1)
    ex, x3 = _x3_and_exact(lambda: ops.pointwise(xi, wi), form)
    assert torch.equal(x3.double(), want) and torch.equal(ex, x3)
    # two sources along K (a stage's first bottleneck: last 1x1 + strided convolutional shortcut as one contraction)
    x1 = torch.randn((2, 13, 21, 128), device='cuda', generator=g)
    x2 = torch.randn((2, 25, 42, 256), device='cuda', generator=g)
    w = torch.randn((512, 128 + 256), device='cuda', generator=g) * 0.05
    b = torch.randn(512, device='cuda', generator=g)
    want = torch.relu(x1.double() @ w[:, :128].double().t() + x2[:, ::2, ::2].double() @ w[:, 128:].double().t() + b.double())
    close(lambda: ops.pointwise_dual(x1, x2, w, b, 2, True), want, 'dual')
    # the lateral convolution with the top-down merge in its epilogue: the merge arithmetic is the float32 kernel's
    c = torch.randn((2, 26, 42, 512), device='cuda', generator=g)
    top = torch.randn((2, 13, 21, 256), device='cuda', generator=g)
    w = torch.randn((256, 512), device='cuda', generator=g) * 0.04
    b = torch.randn(256, device='cuda', generator=g)
    ex, x3 = _x3_and_exact(lambda: ops.lateral_merge(c, w, b, top), form)
    lat = _x3_and_exact(lambda: ops.pointwise(c, w, b), form)[1]
    assert torch.equal(x3, ops.fpn_topdown_merge(top, lat))                       # bit-identical to merge(conv_x3)
    assert float((x3 - ex).abs().max()) <= 1e-5 * float(ex.abs().max())
    # all pyramid levels in one launch == level by level: the same bits without a workspace (no K split anywhere); with it a
    # small level launched alone splits its K loop (another, equally valid summation order): float32 rounding apart
    xs = [torch.randn((2, h, w_, 256), device='cuda', generator=g) for h, w_ in ((48, 64), (24, 32), (12, 16), (6, 8), (3, 4))]
    w = (torch.randn((512, 256, 3, 3), device='cuda', generator=g) * 0.02).contiguous(memory_format=torch.channels_last)
    b = torch.randn(512, device='cuda', generator=g)
    from tf_eager_object_detection_amd import _lib
    from tools._diag import diag_library
    with ops.f32_form(form):
        together = ops.conv3x3_f32_levels(xs, w, b, relu=True)
        single = [ops.conv3x3_f32(x, w, b, relu=True) for x in xs]
        with diag_library():                              # (the -DODET_DIAG build: the shipped library has no tile override)
            _lib.call('odet_debug_x3_tile', 4, 4, 1)                              # (one tile, no split: the launch grouping alone)
            together_1 = ops.conv3x3_f32_levels(xs, w, b, relu=True)
            single_1 = [ops.conv3x3_f32(x, w, b, relu=True) for x in xs]
    assert all(torch.equal(a, s_) for a, s_ in zip(together_1, single_1))
    assert all(float((a - s_).abs().max()) <= 2e-6 * float(s_.abs().max()) for a, s_ in zip(together, single))
    # the limb planes: exact sum (three bfloat16 limbs) / the scaled weight to within one float32 ulp (two float16 limbs: 23 bits)
    wk = w.permute(0, 2, 3, 1).contiguous()
    if form == 'x3':
        planes = ops.split_bf16x3(wk)
        assert torch.equal(planes.view(torch.bfloat16).double().sum(0), wk.double())
    else:
        planes, w_exp = ops.split_f16x2(wk)
        assert 512.0 <= float(wk.abs().max()) * 2.0 ** w_exp < 1024.0
        pl = planes.view(torch.float16).double()
        scaled = wk.double() * 2.0 ** w_exp
        assert bool(((pl[0] + pl[1] * 2.0 ** -11 - scaled).abs() <= 2.0 ** -23 * scaled.abs()).all())     # one float32 ulp


@pytest.mark.gpu
@pytest.mark.parametrize('form', SPLIT_FORMS)
def test_float32_x3_detector_agrees_with_the_exact_float32_detector(form):
    """the split-precision float32 mode against the exact-float32 mode on the SAME weights and images:
    class scores / boxes within 1e-4 (relative to the image scale for boxes), the RPN's kept anchor indices equal up to the
    ties float32 rounding decides (reported; >= 99 % here), same number of detections"""
    from tf_eager_object_detection_amd.model.fpn_detector import ResNetFpnDetector
    torch.manual_seed(1)
    shape, K = (256, 352), 300
    a = ResNetFpnDetector(50, 21, shape, K, dtype=torch.float32, max_batch=2, blind_chunks=3).prepare()
    b = ResNetFpnDetector(50, 21, shape, K, dtype=torch.float32, max_batch=2, blind_chunks=3, f32_form=form)
    b.load_state_dict(a.state_dict())
    b.prepare()
    rng = np.random.default_rng(1)
    img = torch.from_numpy((rng.uniform(0, 255, (2,) + shape + (3,)) - 110).astype(np.float32)).cuda()
    oa, ob = a(img), b(img)
    torch.cuda.synchronize()
    sa, da, _, _ = a._last_pass
    sb, db, _, _ = b._last_pass
    assert float((sa - sb).abs().max()) <= 1e-4 * max(1.0, float(sa.abs().max()))
    assert float((da - db).abs().max()) <= 1e-4 * max(1.0, float(da.abs().max()))
    for i in range(2):
        ka, kb = int(a._hot[i].roi_count.item()), int(b._hot[i].roi_count.item())
        ia, ib = set(a._hot[i].roi_idx[:ka].tolist()), set(b._hot[i].roi_idx[:kb].tolist())
        assert len(ia & ib) >= 0.99 * max(len(ia), 1), (len(ia & ib), len(ia))
        na, nb = int(oa[i][3].item()), int(ob[i][3].item())
        assert abs(na - nb) <= 1 and na > 0


@pytest.mark.gpu
@pytest.mark.parametrize('family', ['fpn', 'vgg16'])
def test_two_limb_pass_out_of_float16_range_is_detected_and_repeated_on_three_limbs(family):
    """f32_form = 'x2' computes on float16 limbs: an activation beyond float16's range makes every sum it enters non-finite and the
    launch's epilogue sets the RANGE STATUS word of the instance's workspace (include/odet.h).  The detectors read that word where
    they read the NMS flags (range_ok) and repeat the pass on the three-limb form (float32's range): images 3000 times brighter than
    any real input give EXACTLY the three-limb detector's results, `range_reruns` counts the pass; ordinary images are not
    re-run.  im_detect (the evaluation entry: ADVICE r5) goes through the same check."""
    from tf_eager_object_detection_amd.model.fpn_detector import ResNetFpnDetector
    from tf_eager_object_detection_amd.model.frcnn_detector import Vgg16Detector
    torch.manual_seed(1)
    shape = (256, 352)
    mk = (lambda form: ResNetFpnDetector(50, 21, shape, 300, dtype=torch.float32, max_batch=2, blind_chunks=3, f32_form=form)) \
        if family == 'fpn' else (lambda form: Vgg16Detector(21, shape, 100, dtype=torch.float32, max_batch=2, f32_form=form))
    a = mk('x3').prepare()
    b = mk('x2')
    b.load_state_dict(a.state_dict())
    b.prepare()
    rng = np.random.default_rng(1)
    img = torch.from_numpy((rng.uniform(0, 255, (2,) + shape + (3,)) - 110).astype(np.float32)).cuda()
    ob = b(img)
    torch.cuda.synchronize()
    assert b.range_reruns == 0 and b.range_ok() and b.f32_form == 'x2'
    n_ok = [int(o[3].item()) for o in ob]
    big = img * 3000.0                                   # (activations of the first layers ~ 1e5 .. 1e6)
    oa = [tuple(t.clone() for t in o) for o in a(big)]
    ob = b(big)
    torch.cuda.synchronize()
    assert b.range_reruns == 1 and b.f32_form == 'x2'
    for x, y in zip(oa, ob):
        n = int(x[3].item())
        assert n == int(y[3].item())
        assert torch.equal(x[0][:n], y[0][:n]) and torch.equal(x[1][:n], y[1][:n]) and torch.equal(x[2][:n], y[2][:n])
    ob = b(img)                                          # back on two limbs
    torch.cuda.synchronize()
    assert b.range_reruns == 1 and [int(o[3].item()) for o in ob] == n_ok
    # the evaluation entry: the same check, the three-limb detector's scores / deltas / rois bit for bit
    ia, ib = a.im_detect(big, 1.0), b.im_detect(big, 1.0)
    assert b.range_reruns == 2 and b.f32_form == 'x2'
    for x, y in zip(ia, ib):
        assert all(torch.equal(u, v) for u, v in zip(x, y)) and bool(torch.isfinite(y[0]).all())
    b.im_detect(img, 1.0)
    assert b.range_reruns == 2


@pytest.mark.gpu
def test_two_limb_overflow_that_a_relu_would_hide_still_sets_the_status_word():
    """ONE activation outside float16's range, all-negative weights, a ReLU behind the layer.  The report does not
    rest on what the layer's OUTPUT looks like after the ReLU (a -inf sum would become 0; with the low limb's -inf the sums are
    in fact NaN here, but nothing downstream is asked to keep them): the epilogue looks at the raw sums and sets the status
    word of the workspace; in-range data and the three-limb form never set it; reading clears it."""
    from tf_eager_object_detection_amd import ops
    g = torch.Generator(device='cuda'); g.manual_seed(5)
    for k in (3, 1):
        x = torch.rand((1, 24, 32, 64), device='cuda', generator=g)
        x[0, 11, 17, 5] = 70000.0                         # the only value beyond 65504
        w = (-torch.rand((128, 64, k, k), device='cuda', generator=g) - 0.01).contiguous(memory_format=torch.channels_last)
        b = torch.zeros(128, device='cuda')
        ws = ops.X3Workspace(x.device)
        run = (lambda: ops.conv3x3_f32(x, w, b, relu=True)) if k == 3 else \
              (lambda: ops.pointwise(x, w.reshape(128, 64).contiguous(), b, None, True))
        with ops.f32_form('x2', workspace=ws):
            y = run()
        y[~torch.isfinite(y)] = 0.0                                               # (whatever a later layer might do to them)
        assert int(ws.range_flag().item()) == 1 and not ws.range_ok() and ws.range_ok()   # reported; reading clears it
        x[0, 11, 17, 5] = 60000.0                                                 # inside the range: no report
        with ops.f32_form('x2', workspace=ws):
            run()
        assert ws.range_ok()
        with ops.f32_form('x3', workspace=ws):                                    # the three-limb form never reports
            x[0, 11, 17, 5] = 70000.0
            run()
        assert ws.range_ok()
    # a workspace is what carries the word: the default one of the stream when the block names none
    w3 = (-torch.rand((128, 64, 3, 3), device='cuda', generator=g) - 0.01).contiguous(memory_format=torch.channels_last)
    ops._x3_workspace(x.device).range_ok()
    with ops.f32_form('x2'):
        ops.conv3x3_f32(x, w3, b, relu=True)
    assert not ops._x3_workspace(x.device).range_ok()


@pytest.mark.gpu
def test_caller_object_on_two_limbs_repeats_an_out_of_range_pass():
    """the reference-surface models take f32_form = 'x2' (VERDICT r5 #8): call() / im_detect on an image with a patch of pixels
    beyond float16's range equal the 'x3' model's results bit for bit, the dense part counts the repeated pass"""
    from tf_eager_object_detection_amd.model.base_fpn_model import ResnetV1Fpn
    torch.manual_seed(3)
    kw = dict(depth=50, rpn_proposal_num_post_nms_test=64, prediction_score_threshold=0.0)
    a = ResnetV1Fpn(f32_form='x3', **kw)
    b = ResnetV1Fpn(f32_form='x2', **kw)
    b._dense_ref.load_state_dict(a._dense_ref.state_dict())
    rng = np.random.default_rng(4)
    img = torch.from_numpy((rng.uniform(0, 255, (1, 192, 256, 3)) - 110).astype(np.float32)).cuda()
    b(img, training=False)
    assert b._dense_ref.range_reruns == 0
    big = img.clone()
    big[0, 50:54, 60:64, :] = 70000.0                    # (a few inputs beyond 65504: everything else stays an ordinary image)
    oa, ob = a(big, training=False), b(big, training=False)
    assert b._dense_ref.range_reruns == 1 and b._dense_ref.f32_form == 'x2'
    for u, v in zip(oa, ob):
        assert (u is None and v is None) or torch.equal(u, v)
    ia, ib = a.im_detect(big, 1.0), b.im_detect(big, 1.0)
    assert b._dense_ref.range_reruns == 2 and all(torch.equal(u, v) for u, v in zip(ia, ib))


@pytest.mark.gpu
def test_float16_detections_of_one_image_do_not_depend_on_the_batch_beyond_rounding():
    """the float16 bottleneck route depends on the batch (the fused 3x3 + 1x1 tail from 200 slabs on, two launches
    below: another accumulation order before the one float16 rounding), so one image's low bits -- and NMS / top-k ties with
    them -- may differ between batch 1 and batch 8.  Documented behaviour; the bound: the same image alone and as image 0 of a
    batch of 8 gives RPN outputs within float16 rounding, >= 95 % of the same kept anchors, and the same detections up to
    that (a labelled box within 2 px and 0.02 in score of a box of the other pass)."""
    from tf_eager_object_detection_amd.model.fpn_detector import ResNetFpnDetector
    torch.manual_seed(1)
    shape, K = (256, 352), 300
    m = ResNetFpnDetector(50, 21, shape, K, dtype=torch.float16, max_batch=8, blind_chunks=3).prepare()
    rng = np.random.default_rng(1)
    img = torch.from_numpy((rng.uniform(0, 255, (8,) + shape + (3,)) - 110).astype(np.float32)).cuda()
    out8 = m(img)
    torch.cuda.synchronize()
    s8, d8 = m._last_pass[0][0].clone(), m._last_pass[1][0].clone()
    idx8 = set(m._hot[0].roi_idx[:int(m._hot[0].roi_count.item())].tolist())
    b8, l8, c8, n8 = [t.clone() for t in out8[0]]
    out1 = m(img[:1])
    torch.cuda.synchronize()
    s1, d1 = m._last_pass[0][0], m._last_pass[1][0]
    idx1 = set(m._hot[0].roi_idx[:int(m._hot[0].roi_count.item())].tolist())
    assert float((s1 - s8).abs().max()) <= 3e-2 * max(1.0, float(s8.abs().max()))
    assert float((d1 - d8).abs().max()) <= 3e-2 * max(1.0, float(d8.abs().max()))
    assert len(idx1 & idx8) >= 0.95 * max(len(idx8), 1), (len(idx1 & idx8), len(idx8))
    b1, l1, c1, n1 = out1[0]
    n1, n8 = int(n1.item()), int(n8.item())
    assert n8 > 0 and abs(n1 - n8) <= max(2, n8 // 10)
    matched = 0
    for i in range(n1):
        same = (l8[:n8] == l1[i]) & ((b8[:n8] - b1[i]).abs().max(dim=1).values <= 2.0) & ((c8[:n8] - c1[i]).abs() <= 0.02)
        matched += int(bool(same.any()))
    assert matched >= 0.9 * n1, (matched, n1)


@pytest.mark.gpu
@pytest.mark.parametrize('form', SPLIT_FORMS)
@pytest.mark.parametrize('kind', ['c4', 'vgg16'])
def test_float32_x3_single_level_detectors_agree_with_the_exact_float32_detectors(kind, form):
    """the split-precision float32 mode of the ResNet-C4 / VGG16 detectors (BASELINE configs 2 / 1) against their exact-float32
    mode on the same weights: RPN outputs within 1e-4, >= 99 % of the same kept anchors, the same number of detections (+-1)"""
    from tf_eager_object_detection_amd.model.frcnn_detector import ResNetC4Detector, Vgg16Detector
    torch.manual_seed(12)
    shape, K = (256, 352), 100
    mk = (lambda **kw: ResNetC4Detector(50, 21, shape, K, dtype=torch.float32, max_batch=2, **kw)) if kind == 'c4' else \
         (lambda **kw: Vgg16Detector(21, shape, K, dtype=torch.float32, max_batch=2, **kw))
    a = mk().prepare()
    b = mk(f32_form=form)
    b.load_state_dict(a.state_dict())
    b.prepare()
    rng = np.random.default_rng(2)
    img = torch.from_numpy((rng.uniform(0, 255, (2,) + shape + (3,)) - 110).astype(np.float32)).cuda()
    oa, ob = a(img), b(img)
    torch.cuda.synchronize()
    sa, da = a._last_pass[0], a._last_pass[1]
    sb, db = b._last_pass[0], b._last_pass[1]
    assert float((sa - sb).abs().max()) <= 1e-4 * max(1.0, float(sa.abs().max()))
    assert float((da - db).abs().max()) <= 1e-4 * max(1.0, float(da.abs().max()))
    for i in range(2):
        ka, kb = int(a._hot[i].roi_count.item()), int(b._hot[i].roi_count.item())
        ia, ib = set(a._hot[i].roi_idx[:ka].tolist()), set(b._hot[i].roi_idx[:kb].tolist())
        assert len(ia & ib) >= 0.99 * max(len(ia), 1), (len(ia & ib), len(ia))
        assert abs(int(oa[i][3].item()) - int(ob[i][3].item())) <= 1


@pytest.mark.gpu
@pytest.mark.parametrize('form', SPLIT_FORMS)
def test_split_precision_split_k_is_deterministic_exact_on_integers_and_leaves_its_workspace_clean(form):
    """the K split of the split-precision kernel (launches with few pixels and deep K: conv5 / the dense layers at small batch):
    S workgroups per tile, float32 parts added in fixed order by the last one.  Forced S = 2, 3, 5, 8 on integer data = the exact
    result; on random data within float32 rounding of the unsplit launch and bit-identical from run to run; the ticket words of
    the workspace are zero after every launch; the product's own pick splits these shapes"""
    from tf_eager_object_detection_amd import ops, _lib
    from tools._diag import diag_library
    g = torch.Generator(device='cuda'); g.manual_seed(77)
    with ops.f32_form(form):
        for (B, H, W, cin, cout, k) in ((1, 25, 42, 512, 512, 3), (1, 1, 1000, 12544, 1024, 1), (2, 13, 21, 2048, 512, 1), (1, 50, 84, 256, 256, 3)):
            xi = torch.randint(-2, 3, (B, H, W, cin), device='cuda', generator=g).float()
            wi = ((torch.randint(0, 100, (cout, cin, k, k), device='cuda', generator=g) < 10).float()
                  * torch.randint(-2, 3, (cout, cin, k, k), device='cuda', generator=g).float()).contiguous(memory_format=torch.channels_last)
            x = torch.randn((B, H, W, cin), device='cuda', generator=g)
            w = (torch.randn((cout, cin, k, k), device='cuda', generator=g) * (cin * k * k) ** -0.5).contiguous(memory_format=torch.channels_last)
            b = torch.randint(-3, 4, (cout,), device='cuda', generator=g).float()
            run = (lambda xx, ww: ops.conv3x3_f32(xx, ww, b, relu=True)) if k == 3 else \
                  (lambda xx, ww: ops.pointwise(xx, ww.reshape(cout, cin).contiguous(), b, None, True))
            want_i = torch.relu(F.conv2d(xi.permute(0, 3, 1, 2).double(), wi.double(), b.double(), 1, k // 2)).permute(0, 2, 3, 1)
            with diag_library():                          # (the -DODET_DIAG build: the shipped library has no tile override)
                _lib.call('odet_debug_x3_tile', 2, 2, 1)
                base = run(x, w)
                for S in (2, 3, 5, 8):
                    _lib.call('odet_debug_x3_tile', 2, 2, S)
                    got_i = run(xi, wi)
                    assert torch.equal(got_i.double(), want_i), (S, B, H, W, cin, cout, k)
                    a1, a2 = run(x, w), run(x, w)
                    assert torch.equal(a1, a2)                                        # deterministic
                    assert float((a1 - base).abs().max()) <= 2e-5 * float(base.abs().max())
            auto = run(x, w)                                                          # the product's own pick (shipped library)
            assert float((auto - base).abs().max()) <= 2e-5 * float(base.abs().max())
        torch.cuda.synchronize()
        assert ops._X3_WS, 'no split-K workspace was allocated'
        for ws in ops._X3_WS.values():
            assert int(ws.buf[:16384].max().item()) == 0                             # every ticket drawn back to zero
            assert ws.range_ok()                                                     # and nothing reported out of range

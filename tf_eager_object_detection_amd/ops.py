"""Functional torch-tensor front-end of the C ABI (one function per odet_* entry point).

Inputs and outputs are GPU tensors; shapes that depend on data (NMS survivors, filters) come
back padded together with a device-side count.  The reference-surface modules in
``model/`` and ``utils/`` slice them to the reference's dynamic shapes (one host sync, as the
reference itself does at region_proposal.py:78 / prediction.py:147); ``pipeline.py`` keeps
everything padded and sync-free.
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib as L

# constants of include/odet.h
RPN_LAYOUT_FPN, RPN_LAYOUT_FRCNN = 0, 1
ROI_NORM_STRIDE, ROI_NORM_IMAGE, ROI_NORM_TP_ALIGN = 0, 1, 2
ROI_POOL_NONE, ROI_POOL_MAX2, ROI_POOL_AVG2 = 0, 1, 2
ASSIGN_MAX_ROIS = 8192
POSTOPS_MAX_ROIS = 4096
POSTOPS_MAX_CANDIDATES = 8192
MAX_LEVELS = 8

# ---- the float32 layers' arithmetic form ------------------------------------------------------------------------------------
# 'exact': v_mfma_f32_16x16x4_f32 (csrc/conv_f32.hip) -- every product and sum rounded to float32 once, a chain of fmaf.
# 'x3':    split precision (csrc/conv_x3.hip) -- float32 operands as three bfloat16 limbs, six limb products per k on the
#          bfloat16 matrix instructions, float32 accumulation: within float32 rounding of the float64 truth like 'exact',
#          2.6 x its peak rate.  Same tensors in memory; only the weights get a cached companion (their limb planes).
# 'x2':    the two-limb float16 form of the same kernel -- h + l * 2^-11 (operands to one float32 ulp), three products, half the
#          matrix work of 'x3'; for data inside float16's RANGE (|activation| <= 65504, else infinities / NaN; include/odet.h).
# A fixed choice of the caller (the detectors' `f32_form` argument), never a timing decision: the two forms round differently.
# The form (and, optionally, the split-K / status workspace the launches of the block use) is AMBIENT state of the calling
# context, held in a contextvars.ContextVar: two Python threads (or asyncio tasks) that enter and leave `f32_form` blocks in any
# interleaving each see their own value -- a form switch is never implicit and can never leak into another thread's layers.
import contextvars

_F32_CTX = contextvars.ContextVar('odet_f32_form', default=('exact', None))


def current_f32_form():
    return _F32_CTX.get()[0]


class f32_form:
    """with ops.f32_form('x3'): ...  -- the form the float32 conv3x3 / pointwise / dense / lateral_merge / pointwise_dual calls
    inside the block run on.  `workspace`: an ops.X3Workspace the block's split-precision launches use (the detectors own one
    per instance); None = the calling stream's default workspace."""

    def __init__(self, form, workspace=None):
        if form not in ('exact', 'x3', 'x2'):
            raise ValueError("f32 form must be 'exact', 'x3' or 'x2'")
        if workspace is not None and not isinstance(workspace, X3Workspace):
            raise TypeError('workspace must be an ops.X3Workspace')
        self.form, self.workspace = form, workspace

    def __enter__(self):
        self._token = _F32_CTX.set((self.form, self.workspace))
        return self

    def __exit__(self, *exc):
        _F32_CTX.reset(self._token)
        return False


def split_bf16x3(w):
    """float32 contiguous GPU tensor (even element count) -> its three bfloat16 limb planes, int16 [3, *w.shape]
    (odet_split_bf16x3): w == plane0 + plane1 + plane2 exactly (round to nearest even at every limb)."""
    if w.dtype != torch.float32 or not w.is_cuda or not w.is_contiguous() or w.numel() % 2:
        raise ValueError('split_bf16x3: needs a contiguous float32 GPU tensor with an even element count')
    planes = torch.empty((3,) + tuple(w.shape), dtype=torch.int16, device=w.device)
    L.call('odet_split_bf16x3', L.dptr(w), C.c_void_p(planes.data_ptr()), w.numel(), L.stream())
    return planes


def f16x2_exponent(w):
    """the power of two the two-limb planes of `w` are scaled by: the largest |w| * 2^e lies in [512, 1024)"""
    top = float(w.abs().max())
    if not math.isfinite(top):
        raise ValueError('split_f16x2: the weights must be finite')
    return 0 if top == 0.0 else max(-100, min(100, 9 - math.frexp(top)[1] + 1))


def split_f16x2(w, w_exp=None):
    """float32 contiguous GPU tensor (even element count) -> (its two float16 limb planes of w * 2^w_exp, int16 [2, *w.shape];
    w_exp) (odet_split_f16x2): w * 2^w_exp = plane0 + plane1 * 2^-11 to within one float32 ulp."""
    if w.dtype != torch.float32 or not w.is_cuda or not w.is_contiguous() or w.numel() % 2:
        raise ValueError('split_f16x2: needs a contiguous float32 GPU tensor with an even element count')
    if w_exp is None:
        w_exp = f16x2_exponent(w)
    planes = torch.empty((2,) + tuple(w.shape), dtype=torch.int16, device=w.device)
    L.call('odet_split_f16x2', L.dptr(w), C.c_void_p(planes.data_ptr()), w.numel(), int(w_exp), L.stream())
    return planes, int(w_exp)


def _plane_key(w):
    """cache key of a weight's limb planes; None (= split again, every time) for tensors without a version counter (inference
    tensors: torch raises on ._version)"""
    try:
        return (w.data_ptr(), w._version, tuple(w.shape))
    except RuntimeError:
        return None


def _x3_planes(holder, w):
    """the limb planes of the (contiguous, float32) weight `w`, kept ON the tensor object the caller passed (`holder`: the
    layer's parameter, or a module's cached concatenation) until that tensor is modified: the planes live exactly as long as
    the weight they belong to -- no global cache that could outlive or be cleared under a captured HIP graph.  A caller that
    passes a fresh temporary every time pays the split every time (correct, slow): keep weights in stable tensors."""
    key = _plane_key(w)
    hit = holder.__dict__.get('_odet_x3') if key is not None else None
    if hit is None or hit[0] != key:
        hit = (key, split_bf16x3(w), w)
        holder.__dict__['_odet_x3'] = hit
    return hit[1]


def _x2_planes(holder, w):
    """the same for the two-limb form: (planes, w_exp) (the exponent is read from the weights -- one host synchronisation per
    weight tensor, at its first use: a warm-up pass before a HIP-graph capture, as for the workspace)"""
    key = _plane_key(w)
    hit = holder.__dict__.get('_odet_x2') if key is not None else None
    if hit is None or hit[0] != key:
        hit = (key, split_f16x2(w), w)
        holder.__dict__['_odet_x2'] = hit
    return hit[1]


class X3Workspace:
    """The caller-owned workspace of the split-precision launches (include/odet.h): split-K ticket words (zero-filled once; every
    launch leaves them zero), the two-limb form's RANGE STATUS word, the split-K parts.  One workspace serves launches that are
    ordered on the device (one stream, or one captured graph replayed by one stream at a time): two launches that run
    CONCURRENTLY must not share one -- the detectors own one per instance for that reason (a captured graph keeps using the
    instance's, whatever stream replays it), bare `ops` calls get one per (device, stream)."""

    def __init__(self, device):
        self.buf = torch.zeros(int(L.lib().odet_x3_workspace_bytes()), dtype=torch.uint8, device=device)
        off = int(L.lib().odet_x2_status_offset())
        self._head = self.buf[:off + 64]
        self._status = self.buf[off:off + 4].view(torch.int32)

    def args(self):
        return C.c_void_p(self.buf.data_ptr()), self.buf.numel()

    def range_flag(self):
        """the status word as a device tensor (int32 [1]; non-zero: a two-limb launch since the last clear saw a non-finite sum)"""
        return self._status

    def range_ok(self, clear=True):
        """True iff no two-limb launch on this workspace since the last clear met an activation outside float16's range (one host
        sync); clears the word when it was set"""
        bad = int(self._status.item()) != 0
        if bad and clear:
            self._status.zero_()
        return not bad

    def reset(self):
        """tickets and status back to zero (after a failed or aborted launch the tickets may be left non-zero)"""
        self._head.zero_()


_X3_WS = {}


def _x3_workspace(device):
    """the workspace of the current context: the one the enclosing f32_form block names, else the CURRENT stream's on `device`
    (allocated at the stream's first split-precision launch -- a warm-up pass before a HIP-graph capture)"""
    ws = _F32_CTX.get()[1]
    if ws is not None:
        return ws
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    ws = _X3_WS.get(key)
    if ws is None:
        ws = X3Workspace(device)
        _X3_WS[key] = ws
    return ws


def invalidate_planes(module_or_tensor):
    """drops the cached limb planes of a weight tensor / of every parameter and cached concatenation of a module: call after
    writing weights in a way that does not bump the tensor version (`.data` writes, set_()); load_state_dict() and the
    detectors' prepare() do.  (In-place ops through the tensor itself bump the version and need nothing.)"""
    import torch.nn as nn
    objs = [module_or_tensor]
    if isinstance(module_or_tensor, nn.Module):
        objs = list(module_or_tensor.parameters()) + list(module_or_tensor.buffers())
        for m in module_or_tensor.modules():
            for v in list(m.__dict__.values()):
                if isinstance(v, torch.Tensor):
                    objs.append(v)
                elif isinstance(v, (tuple, list)):
                    objs += [t for t in v if isinstance(t, torch.Tensor)]
    for t in objs:
        t.__dict__.pop('_odet_x3', None)
        t.__dict__.pop('_odet_x2', None)


def _f32_sym(sym, w, holder=None):
    """(entry point, weight pointer, extra arguments before the stream) of a float32 layer in the current form; `holder` = the
    caller's weight tensor object"""
    form = _F32_CTX.get()[0]
    if form == 'x3':
        planes = _x3_planes(w if holder is None else holder, w)
        return sym[:-3] + 'x3', C.c_void_p(planes.data_ptr()), _x3_workspace(w.device).args()   # odet_*_f32 -> odet_*_x3
    if form == 'x2':
        planes, w_exp = _x2_planes(w if holder is None else holder, w)
        return sym[:-3] + 'x2', C.c_void_p(planes.data_ptr()), (w_exp,) + _x3_workspace(w.device).args()
    return sym, L.dptr(w), ()


def _f32_call(sym, *args):
    """L.call for a float32 layer; a split-precision launch that FAILS may leave its workspace's tickets non-zero: reset them"""
    try:
        L.call(sym, *args)
    except L.OdetError:
        if sym.endswith(('_x3', '_x2', '_x3_levels', '_x2_levels')):
            _x3_workspace(torch.device('cuda', torch.cuda.current_device())).reset()
        raise


def _boxes(t, name):
    t = L.f32c(t, name)
    if t.dim() != 2 or t.shape[1] != 4:
        raise ValueError('%s must have shape [N,4], got %s' % (name, tuple(t.shape)))
    return t


def anchors_shift(anchor_base, feat_stride, height, width):
    base = _boxes(anchor_base, 'anchor_base')
    fh, fw = int(height), int(width)
    out = torch.empty((fh * fw * base.shape[0], 4), dtype=torch.float32, device=base.device)
    L.check(L.lib().odet_anchors_shift(L.dptr(base), base.shape[0], int(feat_stride), fh, fw, L.dptr(out), L.stream()))
    return out


def anchors_fpn(fh_list, fw_list, stride_list, wh_table, device):
    """wh_table: float32 numpy [num_levels, A, 2] of (w, h) per level / anchor."""
    nl = len(fh_list)
    wh = np.ascontiguousarray(wh_table, dtype=np.float32)
    A = wh.shape[1]
    fh = (C.c_int * nl)(*[int(v) for v in fh_list])
    fw = (C.c_int * nl)(*[int(v) for v in fw_list])
    st = (C.c_int * nl)(*[int(v) for v in stride_list])
    total = sum(int(a) * int(b) for a, b in zip(fh_list, fw_list)) * A
    out = torch.empty((total, 4), dtype=torch.float32, device=device)
    L.check(L.lib().odet_anchors_fpn(nl, A, fh, fw, st, wh.ctypes.data_as(C.c_void_p), L.dptr(out), L.stream()))
    return out


def decode(anchors, deltas, means, stds, clip_shape=None, out=None):
    anchors = _boxes(anchors, 'anchors')
    deltas = L.f32c(deltas, 'deltas')
    n = anchors.shape[0]
    if deltas.numel() != n * 4:
        raise ValueError('deltas must hold %d x 4 values, got shape %s' % (n, tuple(deltas.shape)))
    if out is None:
        out = torch.empty((n, 4), dtype=torch.float32, device=anchors.device)
    ch, cw = (0, 0) if clip_shape is None else (int(clip_shape[0]), int(clip_shape[1]))
    L.check(L.lib().odet_decode(L.dptr(anchors), L.dptr(deltas), 4, n, L.host4(means, 'target_means'),
                                L.host4(stds, 'target_stds'), ch, cw, L.dptr(out), L.stream()))
    return out


def encode(src, dst, means, stds):
    src = _boxes(src, 'src_bbox')
    dst = _boxes(dst, 'dst_bbox')
    if src.shape != dst.shape:
        raise ValueError('src_bbox and dst_bbox shapes differ: %s vs %s' % (tuple(src.shape), tuple(dst.shape)))
    out = torch.empty_like(src)
    L.check(L.lib().odet_encode(L.dptr(src), L.dptr(dst), src.shape[0], L.host4(means, 'target_means'),
                                L.host4(stds, 'target_stds'), L.dptr(out), L.stream()))
    return out


def clip(boxes, min_value, max_height, max_width):
    boxes = _boxes(boxes, 'boxes')
    out = torch.empty_like(boxes)
    L.check(L.lib().odet_clip(L.dptr(boxes), boxes.shape[0], float(min_value), int(max_height), int(max_width),
                              L.dptr(out), L.stream()))
    return out


def _count_tensor(device):
    return torch.zeros(1, dtype=torch.int32, device=device)


def clip_filter(boxes, min_value, max_height, max_width, min_edge):
    """-> (boxes [n,4] padded, idx int64 [n] padded, count int32[1] on device)"""
    boxes = _boxes(boxes, 'boxes')
    n = boxes.shape[0]
    ob = torch.empty_like(boxes)
    oi = torch.empty(n, dtype=torch.int64, device=boxes.device)
    cnt = _count_tensor(boxes.device)
    nb = L.lib().odet_compact_workspace_bytes(n)
    ws = L.workspace(nb, boxes.device)
    L.check(L.lib().odet_clip_filter(L.dptr(boxes), n, float(min_value), int(max_height), int(max_width),
                                     float(min_edge), L.dptr(ob), L.dptr(oi), L.dptr(cnt), L.dptr(ws), nb, L.stream()))
    return ob, oi, cnt


def range_filter(boxes, max_height, max_width):
    boxes = _boxes(boxes, 'anchors')
    n = boxes.shape[0]
    oi = torch.empty(n, dtype=torch.int64, device=boxes.device)
    cnt = _count_tensor(boxes.device)
    nb = L.lib().odet_compact_workspace_bytes(n)
    ws = L.workspace(nb, boxes.device)
    L.check(L.lib().odet_range_filter(L.dptr(boxes), n, int(max_height), int(max_width), L.dptr(oi), L.dptr(cnt),
                                      L.dptr(ws), nb, L.stream()))
    return oi, cnt


def where_greater(values, threshold):
    """tf.where(values > threshold)[:, 0] for a 1-D (possibly strided) float32 GPU tensor."""
    if values.dim() != 1:
        raise ValueError('values must be 1-D')
    if values.dtype != torch.float32 or not values.is_cuda:
        values = L.f32c(values, 'values')
    n = values.shape[0]
    stride = values.stride(0) if n > 1 else 1
    if stride < 1:
        values = values.contiguous()
        stride = 1
    oi = torch.empty(n, dtype=torch.int64, device=values.device)
    cnt = _count_tensor(values.device)
    nb = L.lib().odet_compact_workspace_bytes(n)
    ws = L.workspace(nb, values.device)
    L.check(L.lib().odet_where_greater(C.c_void_p(values.data_ptr()), stride, n, float(threshold), L.dptr(oi),
                                       L.dptr(cnt), L.dptr(ws), nb, L.stream()))
    return oi, cnt


def pairwise_iou(b1, b2):
    b1 = _boxes(b1, 'boxlist1')
    b2 = _boxes(b2, 'boxlist2')
    out = torch.empty((b1.shape[0], b2.shape[0]), dtype=torch.float32, device=b1.device)
    L.check(L.lib().odet_pairwise_iou(L.dptr(b1), b1.shape[0], L.dptr(b2), b2.shape[0], L.dptr(out), L.stream()))
    return out


def gather_rows(src, idx, count_dev=None):
    src = L.f32c(src, 'src')
    row = int(np.prod(src.shape[1:])) if src.dim() > 1 else 1
    if idx.dtype not in (torch.int32, torch.int64):
        raise TypeError('idx must be int32 or int64')
    idx = idx.contiguous()
    n = idx.shape[0]
    out = torch.empty((n,) + tuple(src.shape[1:]), dtype=torch.float32, device=src.device)
    L.check(L.lib().odet_gather_rows(L.dptr(src), L.dptr(idx), 1 if idx.dtype == torch.int64 else 0, n,
                                     L.dptr(count_dev), row, L.dptr(out), L.stream()))
    return out


def rpn_fg_softmax(logits, num_anchors, layout):
    logits = L.f32c(logits, 'rpn_score')
    if layout == RPN_LAYOUT_FPN:
        if logits.numel() % 2:
            raise ValueError('FPN rpn scores must reshape to [-1, 2]')
        n = logits.numel() // 2
        out = torch.empty(n, dtype=torch.float32, device=logits.device)
        L.check(L.lib().odet_rpn_fg_softmax(L.dptr(logits), n, 1, layout, L.dptr(out), L.stream()))
    else:
        A = int(num_anchors)
        if logits.numel() % (2 * A):
            raise ValueError('rpn scores must reshape to [-1, 2*num_anchors]')
        nloc = logits.numel() // (2 * A)
        out = torch.empty(nloc * A, dtype=torch.float32, device=logits.device)
        L.check(L.lib().odet_rpn_fg_softmax(L.dptr(logits), nloc, A, layout, L.dptr(out), L.stream()))
    return out


def nms(boxes, scores, max_output_size, iou_threshold, with_boxes=False, blind_chunks=1, done=None):
    """-> (idx int32 [K] padded, count int32[1]) (+ gathered boxes [K,4]).  ``done`` (int32[1]
    GPU tensor) switches to the sync-free mode of odet_nms."""
    boxes = _boxes(boxes, 'boxes')
    scores = L.f32c(scores, 'scores').reshape(-1)
    n = boxes.shape[0]
    if scores.shape[0] != n:
        raise ValueError('scores has %d entries for %d boxes' % (scores.shape[0], n))
    K = max(min(int(max_output_size), n), 0)
    idx = torch.empty(max(K, 1), dtype=torch.int32, device=boxes.device)
    ob = torch.empty((max(K, 1), 4), dtype=torch.float32, device=boxes.device) if with_boxes else None
    cnt = _count_tensor(boxes.device)
    nb = L.lib().odet_nms_workspace_bytes(n, K)
    ws = L.workspace(nb, boxes.device)
    L.check(L.lib().odet_nms(L.dptr(boxes), L.dptr(scores), n, K, float(iou_threshold), L.dptr(idx), L.dptr(ob),
                             L.dptr(cnt), int(blind_chunks), L.dptr(done), L.dptr(ws), nb, L.stream()))
    return (idx, cnt, ob) if with_boxes else (idx, cnt)


def region_proposal(deltas, anchors, scores, image_shape, num_post_nms, iou_threshold, means, stds,
                    workspace=None, blind_chunks=1, done=None, out=None):
    """-> (rois [K,4] padded, idx int32 [K] padded, count int32[1]).  ``done`` (int32[1] GPU
    tensor) switches to the sync-free mode of odet_region_proposal."""
    anchors = _boxes(anchors, 'anchors')
    deltas = L.f32c(deltas, 'bboxes_txtytwth')
    scores = L.f32c(scores, 'scores').reshape(-1)
    n = anchors.shape[0]
    if deltas.numel() != n * 4 or scores.shape[0] != n:
        # the reference would raise InvalidArgumentError from TF on mismatched N
        raise ValueError('RegionProposal: %d anchors but deltas %s / scores %s'
                         % (n, tuple(deltas.shape), tuple(scores.shape)))
    K = max(min(int(num_post_nms), n), 0)
    if out is not None:
        rois, idx, cnt = out
    else:
        rois = torch.empty((max(K, 1), 4), dtype=torch.float32, device=anchors.device)
        idx = torch.empty(max(K, 1), dtype=torch.int32, device=anchors.device)
        cnt = _count_tensor(anchors.device)
    nb = L.lib().odet_region_proposal_workspace_bytes(n, K)
    ws = workspace if workspace is not None and workspace.numel() >= nb else L.workspace(nb, anchors.device)
    L.check(L.lib().odet_region_proposal(L.dptr(deltas), L.dptr(anchors), L.dptr(scores), n, int(image_shape[0]),
                                         int(image_shape[1]), L.host4(means, 'target_means'),
                                         L.host4(stds, 'target_stds'), K, float(iou_threshold), L.dptr(rois),
                                         L.dptr(idx), L.dptr(cnt), int(blind_chunks), L.dptr(done), L.dptr(ws),
                                         ws.numel(), L.stream()))
    return rois, idx, cnt


FUSED_ORDER_MAX_ROIS = 1024          # ODET_FUSED_ORDER_MAX_ROIS


def fpn_proposals(rpn_logits, rpn_deltas, fh_list, fw_list, stride_list, wh_table, image_shape, num_post_nms,
                  iou_threshold, means, stds, min_level=None, max_level=None, workspace=None, blind_chunks=1,
                  done=None, out=None, out_levels=None, out_order=None):
    """The whole FPN proposal stage (anchors in registers -> fg softmax -> decode+clip -> NMS over all
    anchors [-> level assignment]) as ONE C-ABI call.  -> (rois [K,4] padded, idx int32 [K], count int32[1])
    and, when min_level is given, (sorted rois [K,4], level int32 [K], perm int64 [K], counts int32 [L]).
    out_order (int32 [K], K <= FUSED_ORDER_MAX_ROIS, needs the level outputs): receives the spatial processing order
    of the level-sorted RoIs (what roi_order computes) from the same launches."""
    logits = L.f32c(rpn_logits, 'rpn_score')
    deltas = L.f32c(rpn_deltas, 'rpn_bbox_txtytwth')
    nl = len(fh_list)
    wh = np.ascontiguousarray(wh_table, dtype=np.float32)
    A = wh.shape[1]
    n = sum(int(a) * int(b) for a, b in zip(fh_list, fw_list)) * A
    if logits.numel() != n * 2 or deltas.numel() != n * 4:
        raise ValueError('fpn_proposals: %d anchors but rpn scores %s / deltas %s'
                         % (n, tuple(logits.shape), tuple(deltas.shape)))
    fh = (C.c_int * nl)(*[int(v) for v in fh_list])
    fw = (C.c_int * nl)(*[int(v) for v in fw_list])
    st = (C.c_int * nl)(*[int(v) for v in stride_list])
    K = max(min(int(num_post_nms), n), 1)
    dev = logits.device
    if out is not None:
        rois, idx, cnt = out
    else:
        rois = torch.empty((K, 4), dtype=torch.float32, device=dev)
        idx = torch.empty(K, dtype=torch.int32, device=dev)
        cnt = _count_tensor(dev)
    lv = None
    if min_level is not None:
        nlv = int(max_level) - int(min_level) + 1
        lv = out_levels if out_levels is not None else (
            torch.empty((K, 4), dtype=torch.float32, device=dev), torch.empty(K, dtype=torch.int32, device=dev),
            torch.empty(K, dtype=torch.int64, device=dev), torch.empty(nlv, dtype=torch.int32, device=dev))
    nb = L.lib().odet_fpn_proposals_workspace_bytes(n, K)
    ws = workspace if workspace is not None and workspace.numel() >= nb else L.workspace(nb, dev)
    L.call(
        'odet_fpn_proposals', L.dptr(logits), L.dptr(deltas), nl, A, fh, fw, st, wh.ctypes.data_as(C.c_void_p), int(image_shape[0]),
        int(image_shape[1]), L.host4(means, 'target_means'), L.host4(stds, 'target_stds'), K, float(iou_threshold),
        int(min_level or 0), int(max_level or 0), L.dptr(rois), L.dptr(idx), L.dptr(cnt),
        L.dptr(lv[0]) if lv else None, L.dptr(lv[1]) if lv else None, L.dptr(lv[2]) if lv else None,
        L.dptr(lv[3]) if lv else None, L.dptr(out_order, torch.int32, 'out_order') if out_order is not None else None,
        int(blind_chunks), L.dptr(done), L.dptr(ws), ws.numel(), L.stream())
    return (rois, idx, cnt) + ((lv,) if lv else ())


def frcnn_proposals(rpn_logits, rpn_deltas, anchor_base, feat_stride, fh, fw, image_shape, num_post_nms,
                    iou_threshold, means, stds, workspace=None, blind_chunks=1, done=None, out=None):
    """The proposal stage of BaseFasterRcnn.call (anchors in registers -> fg softmax of the [A bg | A fg]
    layout -> decode+clip -> NMS over all anchors) as ONE C-ABI call.
    -> (rois [K,4] padded, idx int32 [K], count int32[1])."""
    logits = L.f32c(rpn_logits, 'rpn_score')
    deltas = L.f32c(rpn_deltas, 'rpn_bbox_txtytwth')
    base = np.ascontiguousarray(np.asarray(anchor_base, dtype=np.float32).reshape(-1, 4))
    A = base.shape[0]
    fh, fw = int(fh), int(fw)
    n = fh * fw * A
    if logits.numel() != n * 2 or deltas.numel() != n * 4:
        raise ValueError('frcnn_proposals: %d anchors but rpn scores %s / deltas %s'
                         % (n, tuple(logits.shape), tuple(deltas.shape)))
    K = max(min(int(num_post_nms), n), 1)
    dev = logits.device
    if out is not None:
        rois, idx, cnt = out
    else:
        rois = torch.empty((K, 4), dtype=torch.float32, device=dev)
        idx = torch.empty(K, dtype=torch.int32, device=dev)
        cnt = _count_tensor(dev)
    nb = L.lib().odet_frcnn_proposals_workspace_bytes(n, K)
    ws = workspace if workspace is not None and workspace.numel() >= nb else L.workspace(nb, dev)
    L.call('odet_frcnn_proposals', L.dptr(logits), L.dptr(deltas), base.ctypes.data_as(C.c_void_p), A,
           int(feat_stride), fh, fw, int(image_shape[0]), int(image_shape[1]), L.host4(means, 'target_means'),
           L.host4(stds, 'target_stds'), K, float(iou_threshold), L.dptr(rois), L.dptr(idx), L.dptr(cnt),
           int(blind_chunks), L.dptr(done), L.dptr(ws), ws.numel(), L.stream())
    return rois, idx, cnt


def assign_levels(rois, min_level, max_level, count_dev=None, out=None):
    """-> (sorted rois [n,4], level int32 [n] (0-based), perm int64 [n], counts int32 [L]).
    ``out`` = preallocated (sorted rois, level, perm, counts) to reuse."""
    rois = _boxes(rois, 'all_rois')
    n = rois.shape[0]
    nl = int(max_level) - int(min_level) + 1
    if out is not None:
        out, lvl, perm, counts = out
    else:
        out = torch.empty_like(rois)
        lvl = torch.empty(max(n, 1), dtype=torch.int32, device=rois.device)
        perm = torch.empty(max(n, 1), dtype=torch.int64, device=rois.device)
        counts = torch.empty(nl, dtype=torch.int32, device=rois.device)
    L.check(L.lib().odet_assign_levels(L.dptr(rois), n, L.dptr(count_dev), int(min_level), int(max_level),
                                       L.dptr(out), L.dptr(lvl), L.dptr(perm), L.dptr(counts), L.stream()))
    return out, lvl[:n], perm[:n], counts


def fpn_topdown_merge(top, lateral, out=None):
    """P_k = 0.5 * tf.image.resize_bilinear(P_{k+1}, size(lateral)) + 0.5 * lateral (reference
    model/fpn/resnet_fpn.py:385-398, TF 1.x legacy resize) in one launch.  ``top`` [B,h,w,C] and
    ``lateral`` [B,H,W,C]: NHWC GPU tensors of the same dtype (float32 or float16); contiguous NHWC memory."""
    if top.dim() != 4 or lateral.dim() != 4:
        raise ValueError('top and lateral must be [B,h,w,C] tensors')
    if top.dtype != lateral.dtype or top.dtype not in (torch.float32, torch.float16):
        raise TypeError('top and lateral must both be float32 or both float16')
    if top.shape[0] != lateral.shape[0] or top.shape[3] != lateral.shape[3]:
        raise ValueError('top %s and lateral %s must agree in batch and channels' % (tuple(top.shape), tuple(lateral.shape)))
    if not top.is_contiguous():
        top = top.contiguous()
    if not lateral.is_contiguous():
        lateral = lateral.contiguous()
    B, h, w, Cc = (int(v) for v in top.shape)
    H, W = int(lateral.shape[1]), int(lateral.shape[2])
    if out is None:
        out = torch.empty_like(lateral)
    L.call('odet_fpn_topdown_merge', L.dptr(top), h, w, L.dptr(lateral), H, W, B, Cc, L.dptr(out),
           1 if top.dtype == torch.float16 else 0, L.stream())
    return out


def bias_act_(x, bias, residual=None, relu=True):
    """In place on an NHWC (or any [..., C] contiguous) activation: x = relu?((x + bias) (+ residual)) -- the
    convolution epilogue of the reference's ResNet blocks / RPN head (resnet_fpn.py:154-205) in one pass."""
    if x.dtype not in (torch.float32, torch.float16) or bias.dtype != x.dtype:
        raise TypeError('x and bias must both be float32 or both float16')
    if residual is not None and (residual.dtype != x.dtype or residual.shape != x.shape):
        raise ValueError('residual must match x in dtype and shape')
    Cc = int(x.shape[-1])
    if bias.numel() != Cc:
        raise ValueError('bias must have %d elements' % Cc)
    L.call('odet_bias_act', L.dptr(x), L.dptr(bias), L.dptr(residual), x.numel() // Cc, Cc, 1 if relu else 0,
           1 if x.dtype == torch.float16 else 0, L.stream())
    return x


def rpn_pack(level_out, bias, out, out_offset):
    """One level of the RPN head's output into the concatenated arrays: ``level_out`` [B,h,w,ch] NHWC (float32 /
    float16, the 1x1 convolution WITHOUT its bias), ``bias`` [ch] of the same dtype, ``out`` float32 [B, N, 2 or 4]
    contiguous; ``out_offset`` = the level's first VALUE (anchor offset x 2 or 4) inside an image."""
    if level_out.dim() != 4 or not level_out.is_contiguous():
        raise ValueError('level_out must be a contiguous NHWC [B,h,w,ch] tensor')
    if level_out.dtype not in (torch.float32, torch.float16) or bias.dtype != level_out.dtype:
        raise TypeError('level_out and bias must both be float32 or both float16')
    B, h, w, ch = (int(v) for v in level_out.shape)
    if out.dtype != torch.float32 or not out.is_contiguous() or out.shape[0] != B:
        raise ValueError('out must be a contiguous float32 [B, N, k] tensor')
    stride = out.numel() // max(B, 1)
    L.call('odet_rpn_pack', L.dptr(level_out), L.dptr(bias), h * w, ch, B, L.dptr(out), stride, int(out_offset),
           1 if level_out.dtype == torch.float16 else 0, L.stream())
    return out


def bias_relu_maxpool(x, bias, kernel, stride, pad=0, ceil_mode=False):
    """maxpool(relu(x + bias)) in one pass: ``x`` NHWC [B,H,W,C] contiguous (float32 / float16; a convolution WITHOUT
    its bias), ``bias`` [C]; window ``kernel`` x ``kernel``, ``stride``, ``pad`` skipped taps on every side;
    ``ceil_mode`` as torch's max_pool2d (TF 'same' pooling for kernel == stride).  Returns NHWC [B,OH,OW,C]."""
    if x.dim() != 4 or not x.is_cuda or not x.is_contiguous() or x.dtype not in (torch.float32, torch.float16):
        raise ValueError('x must be a contiguous NHWC float32 / float16 GPU tensor')
    B, H, W, Cn = (int(v) for v in x.shape)
    if bias.dtype != x.dtype or bias.numel() != Cn or not bias.is_contiguous():
        raise ValueError('bias must be a contiguous [C] tensor of the same dtype')

    def osz(n):
        a = n + 2 * pad - kernel
        o = (-(-a // stride) if ceil_mode else a // stride) + 1
        if ceil_mode and (o - 1) * stride >= n + pad:        # torch: the last window must start inside the map
            o -= 1
        return o
    OH, OW = osz(H), osz(W)
    out = torch.empty((B, OH, OW, Cn), dtype=x.dtype, device=x.device)
    L.call('odet_bias_relu_maxpool', L.dptr(x), L.dptr(bias), L.dptr(out), B, H, W, Cn, OH, OW, int(kernel), int(stride),
           int(pad), 1 if x.dtype == torch.float16 else 0, L.stream())
    return out


def rpn_head_tail(conv_out, conv_bias, weight, bias, num_anchors, scores, deltas, anchor_offset):
    """Everything of the RpnHead after its 3x3 convolution for one level, in one MFMA pass: ``conv_out`` [B,h,w,512]
    float16 NHWC contiguous (the 3x3 convolution WITHOUT its bias), ``conv_bias`` [512], ``weight`` [6A,512(,1,1)]
    (2A score rows then 4A delta rows), ``bias`` [6A]; relu(conv_out + conv_bias) . weight^T + bias -> the level's
    slices of ``scores`` [B,N,2] / ``deltas`` [B,N,4] (float32) starting at anchor ``anchor_offset``."""
    A = int(num_anchors)
    if conv_out.dim() != 4 or conv_out.dtype != torch.float16 or not conv_out.is_contiguous() or conv_out.shape[3] != 512:
        raise ValueError('conv_out must be a contiguous float16 NHWC [B,h,w,512] tensor')
    if weight.dtype != torch.float16 or weight.numel() != 6 * A * 512:
        raise ValueError('weight must be a float16 [6A, 512] tensor')
    w = weight.reshape(6 * A, 512)
    if not w.is_contiguous():
        w = w.contiguous()
    for t, n_ in ((conv_bias, 512), (bias, 6 * A)):
        if t.dtype != torch.float16 or t.numel() != n_ or not t.is_contiguous():
            raise ValueError('conv_bias [512] / bias [6A] must be contiguous float16 tensors')
    B, h, w_ = (int(v) for v in conv_out.shape[:3])
    for t, k in ((scores, 2), (deltas, 4)):
        if t.dtype != torch.float32 or not t.is_contiguous() or t.dim() != 3 or t.shape[0] != B or t.shape[2] != k:
            raise ValueError('scores / deltas must be contiguous float32 [B, N, 2] / [B, N, 4] tensors')
    L.call('odet_rpn_head_tail_f16', L.dptr(conv_out), L.dptr(conv_bias), L.dptr(w), L.dptr(bias), h * w_, A, B,
           L.dptr(scores), scores.numel() // max(B, 1), int(anchor_offset) * 2, L.dptr(deltas),
           deltas.numel() // max(B, 1), int(anchor_offset) * 4, L.stream())
    return scores, deltas


_CONV3X3_FORMS = {torch.float16: ('odet_conv3x3_f16', 'float16', 64), torch.float32: ('odet_conv3x3_f32', 'float32', 32)}


def _conv3x3_weight(weight, cin, cout, dtype, name):
    if weight.dtype != dtype or weight.dim() != 4 or weight.numel() != cout * 9 * cin:
        raise ValueError('weight must be %s [cout,cin,3,3] (channels_last) or [cout,3,3,cin]' % name)
    if tuple(weight.shape[1:]) == (cin, 3, 3):
        w = weight.permute(0, 2, 3, 1)                    # [cout,3,3,cin] view; contiguous iff channels_last memory
    elif tuple(weight.shape[1:]) == (3, 3, cin):
        w = weight
    else:
        raise ValueError('weight shape %s does not match cin = %d' % (tuple(weight.shape), cin))
    return w if w.is_contiguous() else w.contiguous()


def _conv3x3(dtype, x, weight, bias, relu, out):
    sym, name, _ = _CONV3X3_FORMS[dtype]
    if x.dtype != dtype or not x.is_cuda or x.dim() != 4 or not x.is_contiguous():
        raise ValueError('x must be a contiguous NHWC %s GPU tensor [B,H,W,cin]' % name)
    B, H, W, cin = (int(v) for v in x.shape)
    cout = int(weight.shape[0])
    w = _conv3x3_weight(weight, cin, cout, dtype, name)
    if bias is not None and (bias.dtype != dtype or bias.numel() != cout or not bias.is_contiguous()):
        raise ValueError('bias must be a contiguous %s [cout] tensor' % name)
    shape = (B, H, W, cout)
    if out is None:
        out = torch.empty(shape, dtype=dtype, device=x.device)
    elif out.dtype != dtype or tuple(out.shape) != shape or not out.is_contiguous():
        raise ValueError('out must be a contiguous %s tensor [B,H,W,cout]' % name)
    wp, extra = L.dptr(w), ()
    if dtype == torch.float32:
        sym, wp, extra = _f32_sym(sym, w, weight)
    _f32_call(sym, L.dptr(x), wp, L.dptr(bias) if bias is not None else None, L.dptr(out), B, H, W,
           cin, cout, 1 if relu else 0, *extra, L.stream())
    return out


def _conv3x3_levels(dtype, xs, weight, bias, relu, outs):
    sym, name, _ = _CONV3X3_FORMS[dtype]
    if not 1 <= len(xs) <= MAX_LEVELS:
        raise ValueError('between 1 and %d maps expected' % MAX_LEVELS)
    B, cin = int(xs[0].shape[0]), int(xs[0].shape[3])
    cout = int(weight.shape[0])
    w = _conv3x3_weight(weight, cin, cout, dtype, name)
    if bias is not None and (bias.dtype != dtype or bias.numel() != cout or not bias.is_contiguous()):
        raise ValueError('bias must be a contiguous %s [cout] tensor' % name)
    lv = (L.OdetConvLevel * len(xs))()
    if outs is None:
        outs = [torch.empty(tuple(x.shape[:3]) + (cout,), dtype=dtype, device=x.device) for x in xs]
    for i, (x, y) in enumerate(zip(xs, outs)):
        if x.dtype != dtype or not x.is_cuda or x.dim() != 4 or not x.is_contiguous() or int(x.shape[0]) != B \
                or int(x.shape[3]) != cin:
            raise ValueError('maps must be contiguous NHWC %s GPU tensors [B,H,W,cin] of one batch size' % name)
        if y.dtype != dtype or tuple(y.shape) != tuple(x.shape[:3]) + (cout,) or not y.is_contiguous():
            raise ValueError('outs must be contiguous %s tensors [B,H,W,cout]' % name)
        lv[i].x, lv[i].y, lv[i].H, lv[i].W = x.data_ptr(), y.data_ptr(), int(x.shape[1]), int(x.shape[2])
    wp, extra = L.dptr(w), ()
    if dtype == torch.float32:
        sym, wp, extra = _f32_sym(sym, w, weight)
    _f32_call(sym + '_levels', lv, len(xs), wp, L.dptr(bias) if bias is not None else None, B, cin, cout,
           1 if relu else 0, *extra, L.stream())
    return outs


def conv3x3_f16(x, weight, bias=None, relu=False, out=None):
    """3x3 stride-1 'same' convolution as ONE hand-written implicit-GEMM kernel on the matrix cores (the RpnHead's
    convolution, base_fpn_model.py:401-417): ``x`` NHWC float16 contiguous [B,H,W,cin]; ``weight`` float16
    [cout,cin,3,3] in channels_last memory format (= [cout][3][3][cin]: `w.contiguous(memory_format=channels_last)`)
    or an explicit [cout,3,3,cin] tensor; ``bias`` float16 [cout] or None; -> NHWC float16 [B,H,W,cout].
    cin % 64 == 0, cout % 64 == 0 (256-channel tiles; 128 / 64-channel tiles when cout is not a multiple of 256)."""
    return _conv3x3(torch.float16, x, weight, bias, relu, out)


def conv3x3_relu_pool2_f16(x, weight, bias, out=None):
    """max_pool2x2/2 'same' (relu(conv3x3(x) + bias)) in ONE launch (odet_conv3x3_relu_pool2_f16; a VGG16 stage's last
    convolution with its pooling, vgg16_faster_rcnn.py:260-342): operands as conv3x3_f16, -> NHWC float16
    [B, ceil(H/2), ceil(W/2), cout]; the un-pooled map is never written."""
    if x.dtype != torch.float16 or not x.is_cuda or x.dim() != 4 or not x.is_contiguous():
        raise ValueError('x must be a contiguous NHWC float16 GPU tensor [B,H,W,cin]')
    B, H, W, cin = (int(v) for v in x.shape)
    cout = int(weight.shape[0])
    w = _conv3x3_weight(weight, cin, cout, torch.float16, 'float16')
    if bias is None or bias.dtype != torch.float16 or bias.numel() != cout or not bias.is_contiguous():
        raise ValueError('bias must be a contiguous float16 [cout] tensor')
    shape = (B, (H + 1) // 2, (W + 1) // 2, cout)
    if out is None:
        out = torch.empty(shape, dtype=torch.float16, device=x.device)
    elif out.dtype != torch.float16 or tuple(out.shape) != shape or not out.is_contiguous():
        raise ValueError('out must be a contiguous float16 tensor %s' % (shape,))
    L.call('odet_conv3x3_relu_pool2_f16', L.dptr(x), L.dptr(w), L.dptr(bias), L.dptr(out), B, H, W, cin, cout, L.stream())
    return out


def conv3x3_f16_levels(xs, weight, bias=None, relu=False, outs=None):
    """conv3x3_f16 with shared weights over a list of NHWC float16 maps [B,H_l,W_l,cin] (the RpnHead over the pyramid
    levels) in ONE launch; -> list of [B,H_l,W_l,cout]."""
    return _conv3x3_levels(torch.float16, xs, weight, bias, relu, outs)


def rpn_head_fused(xs, conv_weight, conv_bias, weight, bias, num_anchors, scores, deltas):
    """The whole RpnHead over the pyramid levels in ONE launch (odet_rpn_head_fused_f16): ``xs`` NHWC float16 maps
    [B,H_l,W_l,cin] in concatenation order, ``conv_weight`` [cout,cin,3,3] (channels_last) / ``conv_bias`` [cout] of the
    3x3 convolution, ``weight`` [6A,cout(,1,1)] / ``bias`` [6A] = the score rows then the delta rows of the two 1x1
    convolutions; relu(conv + conv_bias) . weight^T + bias -> ``scores`` [B,N,2] / ``deltas`` [B,N,4] float32 (a
    workgroup walks the channel tiles of its pixel slab: no workspace, no second launch).  cout in {256, 512}."""
    A = int(num_anchors)
    if not 1 <= len(xs) <= MAX_LEVELS:
        raise ValueError('between 1 and %d maps expected' % MAX_LEVELS)
    B, cin = int(xs[0].shape[0]), int(xs[0].shape[3])
    cout = int(conv_weight.shape[0])
    w3 = _conv3x3_weight(conv_weight, cin, cout, torch.float16, 'float16')
    if weight.dtype != torch.float16 or weight.numel() != 6 * A * cout:
        raise ValueError('weight must be a float16 [6A, cout] tensor')
    w1 = weight.reshape(6 * A, cout)
    if not w1.is_contiguous():
        w1 = w1.contiguous()
    for t, n_ in ((conv_bias, cout), (bias, 6 * A)):
        if t.dtype != torch.float16 or t.numel() != n_ or not t.is_contiguous():
            raise ValueError('conv_bias [cout] / bias [6A] must be contiguous float16 tensors')
    lv = (L.OdetConvLevel * len(xs))()
    n = 0
    for i, x in enumerate(xs):
        if x.dtype != torch.float16 or not x.is_cuda or x.dim() != 4 or not x.is_contiguous() or int(x.shape[0]) != B \
                or int(x.shape[3]) != cin:
            raise ValueError('maps must be contiguous NHWC float16 GPU tensors [B,H,W,cin] of one batch size')
        lv[i].x, lv[i].y, lv[i].H, lv[i].W = x.data_ptr(), None, int(x.shape[1]), int(x.shape[2])
        n += int(x.shape[1]) * int(x.shape[2]) * A
    for t, k in ((scores, 2), (deltas, 4)):
        if t.dtype != torch.float32 or not t.is_contiguous() or tuple(t.shape) != (B, n, k):
            raise ValueError('scores / deltas must be contiguous float32 [B, N, 2] / [B, N, 4] tensors, N = sum H*W*A')
    L.call('odet_rpn_head_fused_f16', lv, len(xs), L.dptr(w3), L.dptr(conv_bias), L.dptr(w1), L.dptr(bias), A, B, cin, cout,
           L.dptr(scores), n * 2, L.dptr(deltas), n * 4, L.stream())
    return scores, deltas


def conv3x3_conv1x1_f16(x, weight2, bias2, weight3, bias3, residual=None, relu=True, out=None):
    """A bottleneck block's 3x3 convolution and its last 1x1 convolution in ONE launch (odet_conv3x3_conv1x1_f16):
    relu(relu(conv3x3(x, weight2) + bias2) . weight3^T + bias3 + residual).  ``x`` NHWC float16 [B,H,W,cin], ``weight2``
    [cmid,cin,3,3] (channels_last) with cmid = 64, 128 or 256 (ResNet conv2 / conv3 / conv4: one workgroup holds all of them),
    ``weight3`` [n3,cmid(,1,1)], ``residual`` / ``out`` NHWC float16 [B,H,W,n3]."""
    if x.dtype != torch.float16 or not x.is_cuda or x.dim() != 4 or not x.is_contiguous():
        raise ValueError('x must be a contiguous NHWC float16 GPU tensor [B,H,W,cin]')
    B, H, W, cin = (int(v) for v in x.shape)
    cmid = int(weight2.shape[0])
    if cmid not in (64, 128, 256):
        raise ValueError('the 3x3 convolution must have 64, 128 or 256 output channels')
    w2 = _conv3x3_weight(weight2, cin, cmid, torch.float16, 'float16')
    n3 = int(weight3.shape[0])
    if weight3.dtype != torch.float16 or weight3.numel() != n3 * cmid:
        raise ValueError('weight3 must be a float16 [n3, cmid] tensor')
    w3 = weight3.reshape(n3, cmid)
    if not w3.is_contiguous():
        w3 = w3.contiguous()
    for t, n_ in ((bias2, cmid), (bias3, n3)):
        if t.dtype != torch.float16 or t.numel() != n_ or not t.is_contiguous():
            raise ValueError('bias2 [cmid] / bias3 [n3] must be contiguous float16 tensors')
    shape = (B, H, W, n3)
    if residual is not None and (residual.dtype != torch.float16 or tuple(residual.shape) != shape or not residual.is_contiguous()):
        raise ValueError('residual must be a contiguous float16 tensor [B,H,W,n3]')
    if out is None:
        out = torch.empty(shape, dtype=torch.float16, device=x.device)
    elif out.dtype != torch.float16 or tuple(out.shape) != shape or not out.is_contiguous():
        raise ValueError('out must be a contiguous float16 tensor [B,H,W,n3]')
    L.call('odet_bottleneck_tail_f16', L.dptr(x), L.dptr(w2), L.dptr(bias2), L.dptr(w3), L.dptr(bias3),
           L.dptr(residual) if residual is not None else None, L.dptr(out), B, H, W, cin, cmid, n3, 1 if relu else 0,
           L.stream())
    return out


def stem_pack_weights(weight):
    """[64,3,7,7] float16 stem weights (any strides) -> the packed [64*7*32] float16 layout of stem_conv7_pool3."""
    if weight.dtype != torch.float16 or tuple(weight.shape) != (64, 3, 7, 7) or not weight.is_cuda:
        raise ValueError('weight must be a float16 [64,3,7,7] GPU tensor')
    packed = torch.empty(64 * 7 * 32, dtype=torch.float16, device=weight.device)
    so, sc, sy, sx = (int(v) for v in weight.stride())
    L.call('odet_stem_pack_weights_f16', C.c_void_p(weight.data_ptr()), so, sc, sy, sx, L.dptr(packed), L.stream())     # (any strides)
    return packed


def stem_conv7_pool3(images_nhwc, packed_weight, bias, out=None):
    """The ResNet stem in ONE launch (odet_stem_conv7_pool3_f16): pad 3 + 7x7/2 convolution + bias + ReLU + pad 1 + 3x3/2
    max-pooling from the NHWC image [B,H,W,3] (float32 or float16) to NHWC float16 [B,PH,PW,64]."""
    x = images_nhwc
    if x.dtype not in (torch.float32, torch.float16) or not x.is_cuda or x.dim() != 4 or int(x.shape[3]) != 3 or not x.is_contiguous():
        raise ValueError('images must be a contiguous NHWC float32 / float16 GPU tensor [B,H,W,3]')
    if packed_weight.dtype != torch.float16 or packed_weight.numel() != 64 * 7 * 32 or not packed_weight.is_contiguous():
        raise ValueError('packed_weight: the result of stem_pack_weights')
    if bias.dtype != torch.float16 or bias.numel() != 64 or not bias.is_contiguous():
        raise ValueError('bias must be a contiguous float16 [64] tensor')
    B, H, W = int(x.shape[0]), int(x.shape[1]), int(x.shape[2])
    ch, cw = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    shape = (B, (ch - 1) // 2 + 1, (cw - 1) // 2 + 1, 64)
    if out is None:
        out = torch.empty(shape, dtype=torch.float16, device=x.device)
    elif out.dtype != torch.float16 or tuple(out.shape) != shape or not out.is_contiguous():
        raise ValueError('out must be a contiguous float16 tensor %s' % (shape,))
    L.call('odet_stem_conv7_pool3_f16', L.dptr(x), 1 if x.dtype == torch.float16 else 0, L.dptr(packed_weight), L.dptr(bias),
           L.dptr(out), B, H, W, L.stream())
    return out


def conv3x3_rgb_pack_weights(weight):
    """[64,3,3,3] float16 weights of a 3-channel 3x3 convolution (any strides) -> the packed [4*2*64*8] float16 layout of
    conv3x3_rgb."""
    if weight.dtype != torch.float16 or tuple(weight.shape) != (64, 3, 3, 3) or not weight.is_cuda:
        raise ValueError('weight must be a float16 [64,3,3,3] GPU tensor')
    packed = torch.empty(4 * 2 * 64 * 8, dtype=torch.float16, device=weight.device)
    so, sc, sy, sx = (int(v) for v in weight.stride())
    L.call('odet_conv3x3_rgb_pack_weights_f16', C.c_void_p(weight.data_ptr()), so, sc, sy, sx, L.dptr(packed), L.stream())
    return packed


def conv3x3_rgb(images_nhwc, packed_weight, bias, relu=True, out=None):
    """VGG16's first convolution in ONE launch (odet_conv3x3_rgb_f16): Conv2D(64, 3x3, 'same') + bias (+ ReLU) from the NHWC
    image [B,H,W,3] (float32 or float16) to NHWC float16 [B,H,W,64]."""
    x = images_nhwc
    if x.dtype not in (torch.float32, torch.float16) or not x.is_cuda or x.dim() != 4 or int(x.shape[3]) != 3 or not x.is_contiguous():
        raise ValueError('images must be a contiguous NHWC float32 / float16 GPU tensor [B,H,W,3]')
    if packed_weight.dtype != torch.float16 or packed_weight.numel() != 4 * 2 * 64 * 8 or not packed_weight.is_contiguous():
        raise ValueError('packed_weight: the result of conv3x3_rgb_pack_weights')
    if bias.dtype != torch.float16 or bias.numel() != 64 or not bias.is_contiguous():
        raise ValueError('bias must be a contiguous float16 [64] tensor')
    B, H, W = int(x.shape[0]), int(x.shape[1]), int(x.shape[2])
    shape = (B, H, W, 64)
    if out is None:
        out = torch.empty(shape, dtype=torch.float16, device=x.device)
    elif out.dtype != torch.float16 or tuple(out.shape) != shape or not out.is_contiguous():
        raise ValueError('out must be a contiguous float16 tensor %s' % (shape,))
    L.call('odet_conv3x3_rgb_f16', L.dptr(x), 1 if x.dtype == torch.float16 else 0, L.dptr(packed_weight), L.dptr(bias),
           L.dptr(out), B, H, W, 1 if relu else 0, L.stream())
    return out


def conv3x3_f32(x, weight, bias=None, relu=False, out=None):
    """conv3x3_f16 in the reference's precision: float32 operands and result, exact-float32 matrix instructions
    (every product and sum rounded to float32 once, as a chain of fmaf).  cin % 32 == 0, cout % 64 == 0."""
    return _conv3x3(torch.float32, x, weight, bias, relu, out)


def conv3x3_f32_levels(xs, weight, bias=None, relu=False, outs=None):
    """conv3x3_f32 with shared weights over a list of NHWC float32 maps in ONE launch."""
    return _conv3x3_levels(torch.float32, xs, weight, bias, relu, outs)


def conv1x1_f16(x, weight, bias, residual=None, relu=True, out=None, in_bias=None):
    """1x1 stride-1 convolution + bias (+ residual) (+ ReLU) in ONE kernel on the matrix cores: ``x`` [..., cin]
    NHWC float16 contiguous (any leading dims), ``weight`` [cout, cin(, 1, 1)], ``bias`` [cout], ``residual`` /
    ``out`` [..., cout] (``out`` may be ``residual``).  cin in {64, 128, 256, 512}, cout % 64 == 0.  ``in_bias`` [cin]:
    ``x`` is the preceding convolution without its bias and ReLU; relu(x + in_bias) is applied on load."""
    if x.dtype != torch.float16 or not x.is_cuda or not x.is_contiguous():
        raise ValueError('x must be a contiguous float16 GPU tensor [..., cin]')
    cin = int(x.shape[-1])
    cout = int(weight.shape[0])
    if weight.dtype != torch.float16 or weight.numel() != cout * cin:
        raise ValueError('weight must be a float16 [cout, cin] (or [cout, cin, 1, 1]) tensor')
    w = weight.reshape(cout, cin)       # (a channels_last [cout, cin, 1, 1] weight is the same memory: a view)
    if not w.is_contiguous():
        w = w.contiguous()
    if bias.dtype != torch.float16 or bias.numel() != cout or not bias.is_contiguous():
        raise ValueError('bias must be a contiguous float16 [cout] tensor')
    shape = tuple(x.shape[:-1]) + (cout,)
    if residual is not None and (residual.dtype != torch.float16 or tuple(residual.shape) != shape
                                 or not residual.is_contiguous()):
        raise ValueError('residual must be a contiguous float16 tensor shaped like the output')
    if out is None:
        out = torch.empty(shape, dtype=torch.float16, device=x.device)
    elif out.dtype != torch.float16 or tuple(out.shape) != shape or not out.is_contiguous():
        raise ValueError('out must be a contiguous float16 tensor shaped like the output')
    if out.data_ptr() == x.data_ptr():
        raise ValueError('out must not alias x')
    if in_bias is not None and (in_bias.dtype != torch.float16 or in_bias.numel() != cin or not in_bias.is_contiguous()):
        raise ValueError('in_bias must be a contiguous float16 [cin] tensor')
    L.call('odet_conv1x1_f16', L.dptr(x), L.dptr(in_bias) if in_bias is not None else None, L.dptr(w), L.dptr(bias),
           L.dptr(residual) if residual is not None else None,
           L.dptr(out), x.numel() // cin, cin, cout, 1 if relu else 0, L.stream())
    return out


_PW_FORMS = {torch.float16: ('f16', 'float16', 64), torch.float32: ('f32', 'float32', 32)}     # suffix, name, K granule


def _pw_args(x, weight, bias, who, min_k=True):
    if x.dtype not in _PW_FORMS:
        raise ValueError('%s: x must be float16 or float32' % who)
    sfx, name, gran = _PW_FORMS[x.dtype]
    if not x.is_cuda or not x.is_contiguous() or x.dim() != 4:
        raise ValueError('%s: x must be a contiguous %s GPU tensor [batch, H, W, cin]' % (who, name))
    cin, cout = int(x.shape[-1]), int(weight.shape[0])
    if weight.dtype != x.dtype or weight.numel() % cout:
        raise ValueError('%s: weight must be a %s [cout, cin(, 1, 1)] tensor' % (who, name))
    w = weight.reshape(cout, -1)
    if w.shape[1] != cin:             # (the kernels read the weight rows with stride cin)
        raise ValueError('%s: weight has %d input channels, x %d' % (who, w.shape[1], cin))
    if not w.is_contiguous():
        w = w.contiguous()
    if bias is not None and (bias.dtype != x.dtype or bias.numel() != cout or not bias.is_contiguous()):
        raise ValueError('%s: bias must be a contiguous %s [cout] tensor' % (who, name))
    if cin % gran or (min_k and cin < 2 * gran) or cout % 64:
        raise ValueError('%s: cin %d must be a multiple of %d (>= %d), cout %d a multiple of 64' % (who, cin, gran, 2 * gran, cout))
    return w, cin, cout, sfx


def pointwise(x, weight, bias=None, residual=None, relu=False, stride=1, out=None):
    """A 1x1 convolution (stride 1 or 2, 'valid') or dense layer as the LDS-staged GEMM on the matrix cores
    (odet_pointwise_f16 / odet_pointwise_f32 by the dtype of ``x``): ``x`` [B,H,W,cin] NHWC contiguous, ``weight``
    [cout, cin(, 1, 1)], ``residual`` / ``out`` [B, ceil(H/stride), ceil(W/stride), cout];
    relu?(x[:, ::stride, ::stride] . w^T + bias + residual).  float32: exact-float32 matrix instructions."""
    w, cin, cout, sfx = _pw_args(x, weight, bias, 'pointwise')
    B, H, W = int(x.shape[0]), int(x.shape[1]), int(x.shape[2])
    stride = int(stride)
    shape = (B, (H + stride - 1) // stride, (W + stride - 1) // stride, cout)
    if residual is not None and (residual.dtype != x.dtype or tuple(residual.shape) != shape or not residual.is_contiguous()):
        raise ValueError('residual must be a contiguous tensor of the output\'s dtype and shape %s' % (shape,))
    if out is None:
        out = torch.empty(shape, dtype=x.dtype, device=x.device)
    elif out.dtype != x.dtype or tuple(out.shape) != shape or not out.is_contiguous():
        raise ValueError('out must be a contiguous tensor %s of x\'s dtype' % (shape,))
    sym, wp, extra = ('odet_pointwise_' + sfx, L.dptr(w), ()) if sfx != 'f32' else _f32_sym('odet_pointwise_f32', w, weight)
    _f32_call(sym, L.dptr(x), wp, L.dptr(bias) if bias is not None else None,
           L.dptr(residual) if residual is not None else None, L.dptr(out), B, H, W, stride, cin, cout, 1 if relu else 0,
           *extra, L.stream())
    return out


def dense(x, weight, bias=None, relu=False, out=None):
    """keras Dense on the same kernel: ``x`` [rows, cin] contiguous, ``weight`` [cout, cin] -> [rows, cout]."""
    if x.dim() != 2:
        raise ValueError('dense: x must be [rows, cin]')
    y = pointwise(x.view(1, 1, x.shape[0], x.shape[1]), weight, bias, None, relu, 1,
                  None if out is None else out.view(1, 1, out.shape[0], out.shape[1]))
    return y.view(x.shape[0], -1)


def lateral_merge(x, weight, bias, top, out=None):
    """The FPN neck's lateral 1x1 convolution with the top-down merge in its epilogue (odet_lateral_merge_f16 / _f32;
    resnet_fpn.py:385-398): 0.5 * resize_bilinear(top) + 0.5 * (x . w^T + bias); ``x`` [B,H,W,cin], ``top`` [B,h,w,cout]
    NHWC contiguous, float16 or float32."""
    w, cin, cout, sfx = _pw_args(x, weight, bias, 'lateral_merge')
    B, H, W = int(x.shape[0]), int(x.shape[1]), int(x.shape[2])
    if top.dtype != x.dtype or not top.is_cuda or not top.is_contiguous() or top.dim() != 4 \
            or top.shape[0] != B or top.shape[3] != cout:
        raise ValueError('top must be a contiguous GPU tensor [batch, h, w, cout] of x\'s dtype')
    shape = (B, H, W, cout)
    if out is None:
        out = torch.empty(shape, dtype=x.dtype, device=x.device)
    elif out.dtype != x.dtype or tuple(out.shape) != shape or not out.is_contiguous():
        raise ValueError('out must be a contiguous tensor %s of x\'s dtype' % (shape,))
    sym, wp, extra = ('odet_lateral_merge_' + sfx, L.dptr(w), ()) if sfx != 'f32' else _f32_sym('odet_lateral_merge_f32', w, weight)
    _f32_call(sym, L.dptr(x), wp, L.dptr(bias) if bias is not None else None, L.dptr(top),
           int(top.shape[1]), int(top.shape[2]), L.dptr(out), B, H, W, cin, cout, *extra, L.stream())
    return out


def pointwise_dual(x1, x2, weight, bias=None, stride=1, relu=True, out=None):
    """relu?([x1 | x2[:, ::stride, ::stride]] . weight^T + bias) in one contraction (odet_pointwise_dual_f16 / _f32): the
    last 1x1 convolution of a stage's first bottleneck together with its convolutional shortcut.  ``x1`` [B,Ho,Wo,cin1],
    ``x2`` [B,H,W,cin2] NHWC contiguous of one dtype, ``weight`` [cout, cin1 + cin2] contiguous."""
    if x1.dtype != x2.dtype or x1.dtype not in _PW_FORMS:
        raise ValueError('pointwise_dual: x1 and x2 must both be float16 or float32')
    sfx, name, gran = _PW_FORMS[x1.dtype]
    for t, nm in ((x1, 'x1'), (x2, 'x2')):
        if not t.is_cuda or not t.is_contiguous() or t.dim() != 4:
            raise ValueError('pointwise_dual: %s must be a contiguous GPU tensor [batch, H, W, C]' % nm)
    B, H, W, c2 = (int(v) for v in x2.shape)
    stride = int(stride)
    Ho, Wo = (H + stride - 1) // stride, (W + stride - 1) // stride
    c1 = int(x1.shape[3])
    if tuple(x1.shape[:3]) != (B, Ho, Wo):
        raise ValueError('pointwise_dual: x1 %s does not match the strided x2 map (%d, %d, %d)' % (tuple(x1.shape), B, Ho, Wo))
    cout = int(weight.shape[0])
    if weight.dtype != x1.dtype or tuple(weight.shape) != (cout, c1 + c2) or not weight.is_contiguous():
        raise ValueError('pointwise_dual: weight must be a contiguous %s [cout, cin1 + cin2] tensor' % name)
    if bias is not None and (bias.dtype != x1.dtype or bias.numel() != cout or not bias.is_contiguous()):
        raise ValueError('pointwise_dual: bias must be a contiguous %s [cout] tensor' % name)
    if c1 % gran or c2 % gran or cout % 64:
        raise ValueError('pointwise_dual: cin1 / cin2 must be multiples of %d, cout of 64' % gran)
    shape = (B, Ho, Wo, cout)
    if out is None:
        out = torch.empty(shape, dtype=x1.dtype, device=x1.device)
    elif out.dtype != x1.dtype or tuple(out.shape) != shape or not out.is_contiguous():
        raise ValueError('out must be a contiguous tensor %s of x1\'s dtype' % (shape,))
    sym, wp, extra = ('odet_pointwise_dual_' + sfx, L.dptr(weight), ()) if sfx != 'f32' else _f32_sym('odet_pointwise_dual_f32', weight)
    _f32_call(sym, L.dptr(x1), c1, L.dptr(x2), c2, H, W, stride, wp,
           L.dptr(bias) if bias is not None else None, L.dptr(out), B, cout, 1 if relu else 0, *extra, L.stream())
    return out


# (the float16 names the round-3 callers and tests use)
pointwise_f16, dense_f16, lateral_merge_f16, pointwise_dual_f16 = pointwise, dense, lateral_merge, pointwise_dual


def stem_patches_f32(images_nhwc):
    """The stem's patch matrix for the float32 mode (odet_stem_patches_f32): NHWC float32 [B,H,W,3] -> [B, Ho, Wo, 160] with
    row = the zero-padded 7 x 7 x 3 window of conv1_pad + the 7x7 / 2 'valid' convolution in (dy, dx, channel) order,
    Ho = (H - 1) // 2 + 1.  The convolution is then `pointwise` on it (weights [64, 160] in the same order)."""
    x = images_nhwc
    if x.dtype != torch.float32 or not x.is_cuda or not x.is_contiguous() or x.dim() != 4 or x.shape[3] != 3:
        raise ValueError('stem_patches_f32: images must be a contiguous float32 GPU tensor [B, H, W, 3]')
    B, H, W = int(x.shape[0]), int(x.shape[1]), int(x.shape[2])
    out = torch.empty((B, (H - 1) // 2 + 1, (W - 1) // 2 + 1, 160), dtype=torch.float32, device=x.device)
    L.call('odet_stem_patches_f32', L.dptr(x), L.dptr(out), B, H, W, L.stream())
    return out


def rgb_patches3x3_f32(images_nhwc):
    """The patch matrix of a 3x3 'same' convolution on a 3-channel image for the float32 mode (odet_rgb_patches3x3_f32): NHWC
    float32 [B,H,W,3] -> [B,H,W,64], row = the zero-padded 3 x 3 x 3 window in (dy, dx, channel) order + zeros.  The
    convolution is then `pointwise` on it (weights [cout, 64] in the same order)."""
    x = images_nhwc
    if x.dtype != torch.float32 or not x.is_cuda or not x.is_contiguous() or x.dim() != 4 or x.shape[3] != 3:
        raise ValueError('rgb_patches3x3_f32: images must be a contiguous float32 GPU tensor [B, H, W, 3]')
    B, H, W = int(x.shape[0]), int(x.shape[1]), int(x.shape[2])
    out = torch.empty((B, H, W, 64), dtype=torch.float32, device=x.device)
    L.call('odet_rgb_patches3x3_f32', L.dptr(x), L.dptr(out), B, H, W, L.stream())
    return out


def dense_f16_out_f32(x, weight, bias=None, relu=False, out=None):
    """The last dense layer with float32 results (odet_dense_f16_out_f32): ``x`` [rows, cin] / ``weight`` [cout, cin]
    float16 contiguous, ``bias`` [cout] float32 -> float32 [rows, cout]; cout % 64 == 0 (pad the weight rows with zeros)."""
    if x.dtype != torch.float16 or not x.is_cuda or not x.is_contiguous() or x.dim() != 2:
        raise ValueError('dense_f16_out_f32: x must be a contiguous float16 GPU tensor [rows, cin]')
    w, cin, cout, _ = _pw_args(x.view(1, 1, x.shape[0], x.shape[1]), weight, None, 'dense_f16_out_f32')
    if bias is not None and (bias.dtype != torch.float32 or bias.numel() != cout or not bias.is_contiguous()):
        raise ValueError('dense_f16_out_f32: bias must be a contiguous float32 [cout] tensor')
    rows = int(x.shape[0])
    if out is None:
        out = torch.empty((rows, cout), dtype=torch.float32, device=x.device)
    elif out.dtype != torch.float32 or tuple(out.shape) != (rows, cout) or not out.is_contiguous():
        raise ValueError('out must be a contiguous float32 tensor [rows, cout]')
    L.call('odet_dense_f16_out_f32', L.dptr(x), L.dptr(w), L.dptr(bias) if bias is not None else None, L.dptr(out), rows,
           cin, cout, 1 if relu else 0, L.stream())
    return out


def rpn_pack_pair(level_out, bias, num_anchors, scores, deltas, anchor_offset):
    """rpn_pack for the RpnHead's two 1x1 convolutions run as one contraction: ``level_out`` [B,h,w,6A] (2A score
    channels then 4A delta channels, no bias), ``bias`` [6A]; writes the level's slices of ``scores`` [B,N,2] and
    ``deltas`` [B,N,4] (float32) starting at anchor ``anchor_offset``."""
    if level_out.dim() != 4 or not level_out.is_contiguous():
        raise ValueError('level_out must be a contiguous NHWC [B,h,w,6A] tensor')
    if level_out.dtype not in (torch.float32, torch.float16) or bias.dtype != level_out.dtype:
        raise TypeError('level_out and bias must both be float32 or both float16')
    B, h, w, ch = (int(v) for v in level_out.shape)
    A = int(num_anchors)
    if ch != 6 * A or bias.numel() != ch:
        raise ValueError('level_out / bias must have 6 * num_anchors channels')
    for t, k in ((scores, 2), (deltas, 4)):
        if t.dtype != torch.float32 or not t.is_contiguous() or t.dim() != 3 or t.shape[0] != B or t.shape[2] != k:
            raise ValueError('scores / deltas must be contiguous float32 [B, N, 2] / [B, N, 4] tensors')
    L.call('odet_rpn_pack_pair', L.dptr(level_out), L.dptr(bias), h * w, A, B, L.dptr(scores),
           scores.numel() // max(B, 1), int(anchor_offset) * 2, L.dptr(deltas), deltas.numel() // max(B, 1),
           int(anchor_offset) * 4, 1 if level_out.dtype == torch.float16 else 0, L.stream())
    return scores, deltas


class ProfEvent:
    """HIP event for odet_roi_pool_timed (the dispatch's own begin / end timestamps)."""

    def __init__(self):
        self.handle = C.c_void_p()
        L.check(L.lib().odet_prof_event_create(C.byref(self.handle)))

    def elapsed_ms(self, stop):
        ms = C.c_float()
        L.check(L.lib().odet_prof_event_elapsed_ms(self.handle, stop.handle, C.byref(ms)))
        return float(ms.value)

    def __del__(self):
        try:
            if self.handle:
                L.lib().odet_prof_event_destroy(self.handle)
        except Exception:
            pass


def roi_order(rois, roi_level, image_shape, count_dev=None, out=None):
    """Spatial processing order (int32 [n]) of the RoIs for roi_pool(order=...): sorted by (level, y, x)."""
    rois = _boxes(rois, 'rois')
    n = rois.shape[0]
    if out is None:
        out = torch.empty(max(n, 1), dtype=torch.int32, device=rois.device)
    if roi_level is not None and roi_level.dtype != torch.int32:
        roi_level = roi_level.to(torch.int32)
    L.call('odet_roi_order', L.dptr(rois), L.dptr(roi_level), n, L.dptr(count_dev), int(image_shape[0]),
           int(image_shape[1]), L.dptr(out), L.stream())
    return out


def roi_pool(feature_maps, rois, roi_level, norm_mode, pool_size, pool_mode, strides=None, image_shape=None,
             count_dev=None, out=None, events=None, order=None):
    """feature_maps: list of NHWC float32 GPU tensors [1,H,W,C] (one per level).  ``events`` = (start, stop)
    ProfEvent pair attached to the dispatch (profiling); ``order`` = int32 processing order (roi_order)."""
    rois = _boxes(rois, 'rois')
    n = rois.shape[0]
    nl = len(feature_maps)
    if nl < 1 or nl > MAX_LEVELS:
        raise ValueError('between 1 and %d feature maps expected' % MAX_LEVELS)
    levels = (L.OdetLevel * nl)()
    keep = []
    Cc = None
    f16 = all(isinstance(fm, torch.Tensor) and fm.dtype == torch.float16 for fm in feature_maps)
    for i, fm in enumerate(feature_maps):
        if f16:
            if not fm.is_cuda:
                raise L.OdetError('shared_layers must live on the GPU: tf_eager_object_detection_amd has no CPU path')
            fm = fm.contiguous()
        else:
            fm = L.f32c(fm, 'shared_layers')
        if fm.dim() != 4 or fm.shape[0] != 1:
            raise ValueError('feature map must be NHWC with batch 1, got %s' % (tuple(fm.shape),))
        if Cc is None:
            Cc = fm.shape[3]
        elif fm.shape[3] != Cc:
            raise ValueError('all levels must share the channel count')
        keep.append(fm)
        levels[i].data = fm.data_ptr()
        levels[i].H = fm.shape[1]
        levels[i].W = fm.shape[2]
        levels[i].stride = float(strides[i]) if strides is not None else 0.0
    P = int(pool_size)
    if out is None:
        out = torch.empty((n, P, P, Cc), dtype=torch.float16 if f16 else torch.float32, device=rois.device)
    ih, iw = (0, 0) if image_shape is None else (int(image_shape[0]), int(image_shape[1]))
    if roi_level is not None and roi_level.dtype != torch.int32:
        roi_level = roi_level.to(torch.int32)
    if f16:
        if out.dtype != torch.float16:
            raise TypeError('out must be float16 for float16 feature maps')
        args = (levels, nl, Cc, L.dptr(rois), L.dptr(roi_level), n, L.dptr(count_dev),
                L.dptr(order, torch.int32, 'order') if order is not None else None, int(norm_mode), ih, iw, P,
                int(pool_mode), C.c_void_p(out.data_ptr()), L.stream())
        if events is None:
            L.call('odet_roi_pool_f16', *args)
        else:
            L.call('odet_roi_pool_f16_timed', *(args + (events[0].handle, events[1].handle)))
    elif order is not None:
        L.call('odet_roi_pool_ordered', levels, nl, Cc, L.dptr(rois), L.dptr(roi_level), n, L.dptr(count_dev),
               L.dptr(order, torch.int32, 'order'), int(norm_mode), ih, iw, P, int(pool_mode), L.dptr(out), L.stream(),
               events[0].handle if events else None, events[1].handle if events else None)
    elif events is None:
        L.call('odet_roi_pool', levels, nl, Cc, L.dptr(rois), L.dptr(roi_level), n, L.dptr(count_dev),
               int(norm_mode), ih, iw, P, int(pool_mode), L.dptr(out), L.stream())
    else:
        L.call('odet_roi_pool_timed', levels, nl, Cc, L.dptr(rois), L.dptr(roi_level), n, L.dptr(count_dev),
               int(norm_mode), ih, iw, P, int(pool_mode), L.dptr(out), L.stream(), events[0].handle, events[1].handle)
    return out


def post_ops(scores, deltas, rois, image_shape, means, stds, max_per_class, max_per_image, nms_iou_threshold,
             score_threshold, min_edge, num_classes, count_dev=None, out=None, workspace=None, record=None):
    """-> (boxes [M,4], labels int32 [M], scores [M]) padded to max_per_image, count int32[1].
    ``out`` = preallocated (boxes, labels, scores, count); ``workspace`` = reusable uint8 buffer;
    ``record`` = float32 [max_per_image*6+1] GPU tensor that also receives the detection record."""
    scores = L.f32c(scores, 'roi_scores_softmax')
    if scores.dim() != 2:
        raise ValueError('roi_scores_softmax must be [num_rois, num_classes]')
    R, Ccls = scores.shape
    deltas = L.f32c(deltas, 'roi_txtytwth')
    if deltas.numel() != R * Ccls * 4:
        raise ValueError('roi_txtytwth must hold [num_rois, num_classes, 4] values')
    rois = _boxes(rois, 'rois')
    if rois.shape[0] != R:
        raise ValueError('rois has %d rows for %d score rows' % (rois.shape[0], R))
    M = max(int(max_per_image), 1)
    if out is not None:
        ob, ol, os_, cnt = out
    else:
        ob = torch.empty((M, 4), dtype=torch.float32, device=scores.device)
        ol = torch.empty(M, dtype=torch.int32, device=scores.device)
        os_ = torch.empty(M, dtype=torch.float32, device=scores.device)
        cnt = _count_tensor(scores.device)
    nb = L.lib().odet_post_ops_workspace_bytes(int(num_classes), int(max_per_class))
    ws = workspace if workspace is not None and workspace.numel() >= nb else L.workspace(nb, scores.device)
    head = (L.dptr(scores), L.dptr(deltas), L.dptr(rois), R, L.dptr(count_dev), Ccls, int(num_classes),
            int(image_shape[0]), int(image_shape[1]), L.host4(means, 'target_means'), L.host4(stds, 'target_stds'),
            int(max_per_class), int(max_per_image), float(nms_iou_threshold), float(score_threshold),
            float(min_edge), L.dptr(ob), L.dptr(ol), L.dptr(os_), L.dptr(cnt))
    if record is None:
        L.call('odet_post_ops', *head, L.dptr(ws), nb, L.stream())
    else:
        if record.numel() < M * 6 + 1:
            raise ValueError('record must hold max_per_image*6+1 floats')
        L.call('odet_post_ops_record', *head, L.dptr(record, torch.float32, 'record'), L.dptr(ws), nb, L.stream())
    return ob, ol, os_, cnt

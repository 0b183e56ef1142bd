"""Static-shape, sync-free FPN detection hot path (MI355X-first arrangement of the reference's
BaseFPN.call inference branch, model/fpn/base_fpn_model.py:208-276, minus the dense conv parts).

The reference-surface modules return dynamically shaped tensors and therefore sync with the host
once per stage (as the TF-eager reference does).  A production loop does not want that: here all
buffers are allocated once at their maximum size, data-dependent counts stay in device memory and
are consumed by the next kernel, and nothing between the RPN head's output and the final padded
detections touches the host -- so the whole stage sequence can be replayed from a HIP graph.

    stage_proposals : ONE C-ABI call (odet_fpn_proposals): anchors in registers -> fg softmax ->
                      decode+clip -> radix select of the best candidates -> bit-matrix NMS -> level
                      assignment (stable partition) in the NMS tail
    stage_roi       : fused crop_and_resize(14x14)+maxpool over P2..P5, level-sorted RoIs
    stage_detect    : per-class filter/decode/clip/NMS (one workgroup per class) -> top-k merge
"""
import numpy as np
import torch

from . import ops
from . import synthetic as syn
from .utils.anchor_generator import fpn_level_tables


class FpnHotPath:
    def __init__(self, image_shape, num_classes=21, num_proposals=1000, channels=256, pool_size=7,
                 rpn_nms_iou=0.7, rpn_means=(0, 0, 0, 0), rpn_stds=(1, 1, 1, 1),
                 roi_means=(0, 0, 0, 0), roi_stds=(0.1, 0.1, 0.2, 0.2), max_per_class=50, max_per_image=50,
                 nms_iou=0.3, score_threshold=0.0, min_level=2, max_level=5, strides=syn.FPN_STRIDES,
                 base_sizes=syn.FPN_BASE_SIZES, ratios=syn.FPN_RATIOS, scales=syn.FPN_SCALES,
                 blind_chunks=1, device=None, spatial_order=True, feature_dtype=torch.float32, nms_first_chunk=0):
        self.image_shape = [int(image_shape[0]), int(image_shape[1])]
        self.num_classes = num_classes
        self.K = num_proposals
        self.C = channels
        self.P = pool_size
        self.cfg = dict(rpn_nms_iou=rpn_nms_iou, rpn_means=list(rpn_means), rpn_stds=list(rpn_stds),
                        roi_means=list(roi_means), roi_stds=list(roi_stds), max_per_class=max_per_class,
                        max_per_image=max_per_image, nms_iou=nms_iou, score_threshold=score_threshold)
        self.min_level, self.max_level = min_level, max_level
        self.strides, self.base_sizes, self.ratios, self.scales = strides, base_sizes, ratios, scales
        self.blind_chunks = blind_chunks
        # 0 = auto; up to 4096 candidates in the first NMS chunk for score distributions with heavy suppression (the
        # step-descriptor path: FpnStreamPool / odet_fpn_step_enqueue*)
        self.nms_first_chunk = int(nms_first_chunk)
        self.device = device or torch.device('cuda', torch.cuda.current_device())
        self.N = syn.num_fpn_anchors(self.image_shape, strides, len(ratios) * len(scales))
        dev = self.device
        K = self.K
        # persistent buffers (allocated once; 288 GB of HBM makes this a non-issue)
        self.fh, self.fw, self.wh = fpn_level_tables(self.image_shape, strides, base_sizes, scales, ratios)
        nb = ops.L.lib().odet_fpn_proposals_workspace_bytes(self.N, K)
        # zero-filled ONCE and private to this object: the library keeps its header clean between calls
        # (odet_fpn_step_t.ws_rpn_clean), so the step-descriptor path spends no launch on zeroing it
        self.ws_rpn = torch.zeros(nb, dtype=torch.uint8, device=dev)
        self.rois = torch.zeros((K, 4), dtype=torch.float32, device=dev)
        self.roi_idx = torch.zeros(K, dtype=torch.int32, device=dev)
        self.roi_count = torch.zeros(1, dtype=torch.int32, device=dev)
        self.nms_done = torch.zeros(1, dtype=torch.int32, device=dev)
        # float16 maps (BASELINE config 5) give float16 RoI features; everything else stays float32
        self.roi_features = torch.zeros((K, pool_size, pool_size, channels), dtype=feature_dtype, device=dev)
        nl = max_level - min_level + 1
        self.sorted_rois = torch.zeros((K, 4), dtype=torch.float32, device=dev)
        self.roi_level = torch.zeros(K, dtype=torch.int32, device=dev)
        self.roi_perm = torch.zeros(K, dtype=torch.int64, device=dev)
        self.level_counts = torch.zeros(nl, dtype=torch.int32, device=dev)
        self.roi_order = torch.zeros(K, dtype=torch.int32, device=dev) if spatial_order and K <= 8192 else None
        # up to 1024 proposals the proposal stage writes the processing order itself (tail of the NMS walk)
        self._fused_order = self.roi_order is not None and K <= ops.FUSED_ORDER_MAX_ROIS
        M = max(max_per_image, 1)
        self.det_boxes = torch.zeros((M, 4), dtype=torch.float32, device=dev)
        self.det_labels = torch.zeros(M, dtype=torch.int32, device=dev)
        self.det_scores = torch.zeros(M, dtype=torch.float32, device=dev)
        self.det_count = torch.zeros(1, dtype=torch.int32, device=dev)
        self.ws_post = torch.zeros(ops.L.lib().odet_post_ops_workspace_bytes(num_classes, max_per_class),
                                   dtype=torch.uint8, device=dev)
        self.record = torch.zeros(M * 6 + 1, dtype=torch.float32, device=dev)
        self._plans = {}

    # Host-side launch plans: every buffer of this object is persistent, so the marshalled ctypes
    # arguments of a stage only depend on the input tensors' addresses and the stream.  The first call
    # with a given set of inputs goes through ops.* (validation, marshalling) and records the C-ABI
    # calls; later calls replay them (a few microseconds of host time per stage instead of ~50).
    def _run(self, stage, inputs, fn):
        ok = all(t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() for t in inputs)
        # (address AND shape: a differently shaped tensor at a recycled address must not replay stale sizes)
        key = (stage, torch.cuda.current_stream().cuda_stream) + tuple((t.data_ptr(), tuple(t.shape)) for t in inputs)
        plan = self._plans.get(key) if ok else None
        if plan is not None:
            for f, a in plan[0]:
                rc = f(*a)
                if rc != 0:
                    ops.L.check(rc)
            return plan[1]
        with ops.L.recording() as calls:
            out = fn()
        if ok:
            # a plan keeps its inputs alive (their addresses must stay theirs while it can be replayed): only a few per
            # stage, so a caller that feeds fresh tensors every pass does not pin them all
            mine = [k for k in self._plans if k[0] == stage]
            if len(mine) >= 4:
                for k in mine:
                    del self._plans[k]
            self._plans[key] = (calls, out, inputs)
        return out

    # ---- stage 1: RPN outputs -> level-sorted proposals -------------------------------------
    def stage_proposals(self, rpn_logits, rpn_deltas, exact=False):
        """rpn_logits [N,2] (bg,fg) as RpnHead emits them (base_fpn_model.py:429), rpn_deltas [N,4].  exact: the library
        checks the NMS state on the host after every chunk and runs as many chunks as the image needs (region_proposal.py:74
        is always exact) -- the recovery of an image the sync-free plan reported incomplete; sets nms_done = 1."""
        if exact:
            ops.fpn_proposals(
                rpn_logits, rpn_deltas, self.fh, self.fw, self.strides, self.wh, self.image_shape, self.K,
                self.cfg['rpn_nms_iou'], self.cfg['rpn_means'], self.cfg['rpn_stds'],
                min_level=self.min_level, max_level=self.max_level, workspace=self.ws_rpn, blind_chunks=1, done=None,
                out=(self.rois, self.roi_idx, self.roi_count),
                out_levels=(self.sorted_rois, self.roi_level, self.roi_perm, self.level_counts),
                out_order=self.roi_order if self._fused_order else None)
            self.nms_done.fill_(1)
            return self.sorted_rois, self.roi_level, self.roi_count
        self._run('proposals', (rpn_logits, rpn_deltas), lambda: ops.fpn_proposals(
            rpn_logits, rpn_deltas, self.fh, self.fw, self.strides, self.wh, self.image_shape, self.K,
            self.cfg['rpn_nms_iou'], self.cfg['rpn_means'], self.cfg['rpn_stds'],
            min_level=self.min_level, max_level=self.max_level, workspace=self.ws_rpn,
            blind_chunks=self.blind_chunks, done=self.nms_done,
            out=(self.rois, self.roi_idx, self.roi_count),
            out_levels=(self.sorted_rois, self.roi_level, self.roi_perm, self.level_counts),
            out_order=self.roi_order if self._fused_order else None))
        # base_fpn_model.py:220 (_get_anchors), :223 (fg softmax), :224 (RegionProposal), :256 / :303-324
        return self.sorted_rois, self.roi_level, self.roi_count

    # ---- stage 2: RoI features ---------------------------------------------------------------
    def stage_roi(self, p_list, events=None):
        """p_list: P2..P5 NHWC feature maps.  -> [K,P,P,C] (rows >= count are zero).  ``events``: a
        (start, stop) ops.ProfEvent pair attached to the kernel dispatch (bench.py's roofline timing)."""
        nl = self.max_level - self.min_level + 1
        maps = list(p_list[:nl])
        def go(ev=None):
            if self.roi_order is not None and not self._fused_order:
                ops.roi_order(self.sorted_rois, self.roi_level, self.image_shape, count_dev=self.roi_count,
                              out=self.roi_order)
            return ops.roi_pool(maps, self.sorted_rois, self.roi_level, ops.ROI_NORM_IMAGE, self.P,
                                ops.ROI_POOL_MAX2, image_shape=self.image_shape, count_dev=self.roi_count,
                                out=self.roi_features, events=ev, order=self.roi_order)    # :257 / :152-161
        if events is not None:
            return go(events)
        if maps[0].dtype != torch.float32:
            return go()                       # (launch plans are kept for the float32 path only)
        return self._run('roi', tuple(maps), go)

    # ---- stage 3: RoI-head outputs -> detections ---------------------------------------------
    def stage_detect(self, cls_softmax, cls_deltas):
        """cls_softmax [K,Ccls], cls_deltas [K,Ccls,4] for the level-sorted RoIs (rows >= count ignored)."""
        c = self.cfg
        return self._run('detect', (cls_softmax, cls_deltas), lambda: ops.post_ops(
            cls_softmax, cls_deltas, self.sorted_rois, self.image_shape, c['roi_means'], c['roi_stds'],
            c['max_per_class'], c['max_per_image'], c['nms_iou'], c['score_threshold'], 16, self.num_classes,
            count_dev=self.roi_count, out=(self.det_boxes, self.det_labels, self.det_scores, self.det_count),
            workspace=self.ws_post, record=self.record))                                  # :267-275

    def stage_record(self):
        """Fixed-size detection record of this image for the image-parallel all-gather (written by the
        merge launch of stage_detect; layout of parallel.pack_detections)."""
        return self.record

    def step(self, rpn_logits, rpn_deltas, p_list, cls_softmax, cls_deltas):
        """One image through the whole hot path (the RoI head that sits between stage 2 and 3 in the
        model is not part of this path; its outputs are inputs here)."""
        self.stage_proposals(rpn_logits, rpn_deltas)
        feats = self.stage_roi(p_list)
        boxes, labels, scores, count = self.stage_detect(cls_softmax, cls_deltas)
        return feats, boxes, labels, scores, count


class FrcnnHotPath:
    """Static-shape, sync-free single-level Faster R-CNN detection hot path: the inference branch of the
    reference's BaseFasterRcnn.call (model/faster_rcnn/base_faster_rcnn_model.py:126-198) minus the
    dense conv parts -- VGG16 (pool 14x14 + 2x2 max, 512 ch) and ResNet C4 (7x7 crop, 1024 ch).

        stage_proposals : odet_frcnn_proposals (anchors in registers -> fg softmax of [A bg | A fg] ->
                          decode+clip -> radix select -> bit-matrix NMS over ALL anchors)     :139-153
        stage_roi       : RoiPoolingCropAndResize((feat, rois, 16)) fused crop (+ max-pool)       :182
        stage_detect    : post_ops_prediction                                                 :189-197"""

    def __init__(self, image_shape, num_classes=21, num_proposals=300, channels=1024, pool_size=7,
                 max_pooling_flag=False, extractor_stride=16, anchor_base_size=16, ratios=(0.5, 1, 2),
                 scales=(8, 16, 32), rpn_nms_iou=0.7, rpn_means=(0, 0, 0, 0), rpn_stds=(1, 1, 1, 1),
                 roi_means=(0, 0, 0, 0), roi_stds=(0.1, 0.1, 0.2, 0.2), max_per_class=50, max_per_image=50,
                 nms_iou=0.3, score_threshold=0.0, blind_chunks=1, device=None, feature_dtype=torch.float32):
        from .utils.anchor_generator import generate_anchor_base
        import math
        self.image_shape = [int(image_shape[0]), int(image_shape[1])]
        self.num_classes, self.K, self.C, self.P = num_classes, num_proposals, channels, pool_size
        self.max_pooling_flag = bool(max_pooling_flag)
        self.stride = int(extractor_stride)
        self.anchor_base = generate_anchor_base(anchor_base_size, ratios, scales).astype(np.float32)   # :83-84
        self.A = self.anchor_base.shape[0]
        self.fh = int(math.ceil(self.image_shape[0] / self.stride))                                    # :140-141
        self.fw = int(math.ceil(self.image_shape[1] / self.stride))
        self.N = self.fh * self.fw * self.A
        self.cfg = dict(rpn_nms_iou=rpn_nms_iou, rpn_means=list(rpn_means), rpn_stds=list(rpn_stds),
                        roi_means=list(roi_means), roi_stds=list(roi_stds), max_per_class=max_per_class,
                        max_per_image=max_per_image, nms_iou=nms_iou, score_threshold=score_threshold)
        self.blind_chunks = blind_chunks
        dev = self.device = device or torch.device('cuda', torch.cuda.current_device())
        K = self.K
        # zero-filled once and private to this object (odet_fpn_step_t.ws_rpn_clean)
        self.ws_rpn = torch.zeros(ops.L.lib().odet_frcnn_proposals_workspace_bytes(self.N, K), dtype=torch.uint8,
                                  device=dev)
        self.roi_order = torch.zeros(K, dtype=torch.int32, device=dev) if K <= ops.FUSED_ORDER_MAX_ROIS else None
        self.rois = torch.zeros((K, 4), dtype=torch.float32, device=dev)
        self.roi_idx = torch.zeros(K, dtype=torch.int32, device=dev)
        self.roi_count = torch.zeros(1, dtype=torch.int32, device=dev)
        self.nms_done = torch.zeros(1, dtype=torch.int32, device=dev)
        # float16 feature maps (a float16 backbone) are taken as they are by the pooled mode (14x14 + max)
        self.roi_features = torch.zeros((K, pool_size, pool_size, channels), dtype=feature_dtype, device=dev)
        M = max(max_per_image, 1)
        self.det_boxes = torch.zeros((M, 4), dtype=torch.float32, device=dev)
        self.det_labels = torch.zeros(M, dtype=torch.int32, device=dev)
        self.det_scores = torch.zeros(M, dtype=torch.float32, device=dev)
        self.det_count = torch.zeros(1, dtype=torch.int32, device=dev)
        self.ws_post = torch.zeros(ops.L.lib().odet_post_ops_workspace_bytes(num_classes, max_per_class),
                                   dtype=torch.uint8, device=dev)
        self.record = torch.zeros(M * 6 + 1, dtype=torch.float32, device=dev)

    def stage_proposals(self, rpn_logits, rpn_deltas, exact=False):
        """rpn_logits [fh*fw, 2A] as RpnHead emits them (:342-350), rpn_deltas [fh*fw*A, 4].  exact: host-checked chunks until
        the NMS is complete (the recovery of an image the sync-free plan reported incomplete); sets nms_done = 1."""
        c = self.cfg
        ops.frcnn_proposals(rpn_logits, rpn_deltas, self.anchor_base, self.stride, self.fh, self.fw,
                            self.image_shape, self.K, c['rpn_nms_iou'], c['rpn_means'], c['rpn_stds'],
                            workspace=self.ws_rpn, blind_chunks=1 if exact else self.blind_chunks,
                            done=None if exact else self.nms_done, out=(self.rois, self.roi_idx, self.roi_count))
        if exact:
            self.nms_done.fill_(1)
        return self.rois, self.roi_count

    def stage_roi(self, feat):
        mode = ops.ROI_POOL_MAX2 if self.max_pooling_flag else ops.ROI_POOL_NONE
        return ops.roi_pool([feat], self.rois, None, ops.ROI_NORM_STRIDE, self.P, mode, strides=[float(self.stride)],
                            count_dev=self.roi_count, out=self.roi_features)

    def stage_detect(self, cls_softmax, cls_deltas):
        c = self.cfg
        return ops.post_ops(cls_softmax, cls_deltas, self.rois, self.image_shape, c['roi_means'], c['roi_stds'],
                            c['max_per_class'], c['max_per_image'], c['nms_iou'], c['score_threshold'], self.stride,
                            self.num_classes, count_dev=self.roi_count,
                            out=(self.det_boxes, self.det_labels, self.det_scores, self.det_count),
                            workspace=self.ws_post, record=self.record)

    def step(self, rpn_logits, rpn_deltas, feat, cls_softmax, cls_deltas):
        self.stage_proposals(rpn_logits, rpn_deltas)
        feats = self.stage_roi(feat)
        boxes, labels, scores, count = self.stage_detect(cls_softmax, cls_deltas)
        return feats, boxes, labels, scores, count


def _fill_step(st, h, stream_handle, rpn_logits, rpn_deltas, p_list, cls_softmax, cls_deltas):
    """Fills the odet_fpn_step_t `st` of FpnHotPath slot `h` for one image's inputs (float32 contiguous GPU
    tensors; feature maps in the slot's feature dtype).  Returns the tensors the caller must keep alive."""
    nl = h.max_level - h.min_level + 1
    maps = list(p_list[:nl])
    for t in (rpn_logits, rpn_deltas, cls_softmax, cls_deltas):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise ValueError('FpnStreamPool.bind needs float32 contiguous GPU tensors')
    fdt = h.roi_features.dtype                       # float16 slots (feature_dtype) take float16 maps
    for t in maps:
        if not (t.is_cuda and t.dtype == fdt and t.is_contiguous()):
            raise ValueError('FpnStreamPool.bind needs %s contiguous GPU feature maps' % fdt)
    tensors = [rpn_logits, rpn_deltas, cls_softmax, cls_deltas] + maps
    if rpn_logits.numel() != h.N * 2 or rpn_deltas.numel() != h.N * 4:
        raise ValueError('%d anchors expected, got rpn scores %s / deltas %s'
                         % (h.N, tuple(rpn_logits.shape), tuple(rpn_deltas.shape)))
    if cls_softmax.dim() != 2 or cls_softmax.shape[0] != h.K or cls_deltas.numel() != cls_softmax.numel() * 4:
        raise ValueError('class scores must be [%d, Ccls] and deltas [%d, Ccls, 4]' % (h.K, h.K))
    c = h.cfg
    st.image_h, st.image_w = h.image_shape
    st.num_levels, st.A = len(h.fh), h.wh.shape[1]
    for l in range(len(h.fh)):
        st.fh[l], st.fw[l], st.stride[l] = h.fh[l], h.fw[l], int(h.strides[l])
    flat = h.wh.reshape(-1)
    for i in range(flat.shape[0]):
        st.wh[i] = float(flat[i])
    for k in range(4):
        st.rpn_means[k], st.rpn_stds[k] = float(c['rpn_means'][k]), float(c['rpn_stds'][k])
        st.roi_means[k], st.roi_stds[k] = float(c['roi_means'][k]), float(c['roi_stds'][k])
    st.num_proposals, st.rpn_nms_iou = h.K, float(c['rpn_nms_iou'])
    st.min_level, st.max_level, st.blind_chunks = h.min_level, h.max_level, h.blind_chunks
    st.nms_first_chunk = getattr(h, 'nms_first_chunk', 0)
    st.num_maps, st.channels, st.pool_size = nl, h.C, h.P
    st.maps_f16 = 1 if fdt == torch.float16 else 0
    for l, fm in enumerate(maps):
        if fm.dim() != 4 or fm.shape[0] != 1 or fm.shape[3] != h.C:
            raise ValueError('feature maps must be NHWC [1,H,W,%d]' % h.C)
        st.maps[l].data, st.maps[l].H, st.maps[l].W, st.maps[l].stride = fm.data_ptr(), fm.shape[1], fm.shape[2], 0.0
    st.ccls, st.num_classes = cls_softmax.shape[1], h.num_classes
    st.max_per_class, st.max_per_image = c['max_per_class'], c['max_per_image']
    st.nms_iou, st.score_threshold, st.min_edge = float(c['nms_iou']), float(c['score_threshold']), 16.0
    st.rpn_logits, st.rpn_deltas = rpn_logits.data_ptr(), rpn_deltas.data_ptr()
    st.cls_scores, st.cls_deltas = cls_softmax.data_ptr(), cls_deltas.data_ptr()
    st.rois, st.roi_idx, st.roi_count = h.rois.data_ptr(), h.roi_idx.data_ptr(), h.roi_count.data_ptr()
    st.nms_done, st.sorted_rois = h.nms_done.data_ptr(), h.sorted_rois.data_ptr()
    st.roi_level, st.roi_perm = h.roi_level.data_ptr(), h.roi_perm.data_ptr()
    st.level_counts, st.roi_features = h.level_counts.data_ptr(), h.roi_features.data_ptr()
    st.roi_order = h.roi_order.data_ptr() if h.roi_order is not None else None
    st.det_boxes, st.det_labels = h.det_boxes.data_ptr(), h.det_labels.data_ptr()
    st.det_scores, st.det_count, st.record = h.det_scores.data_ptr(), h.det_count.data_ptr(), h.record.data_ptr()
    st.ws_rpn, st.ws_rpn_bytes = h.ws_rpn.data_ptr(), h.ws_rpn.numel()
    st.ws_rpn_clean = 1                              # (FpnHotPath zero-filled it at allocation and never exposes it)
    st.ws_post_clean = 1
    st.ws_post, st.ws_post_bytes = h.ws_post.data_ptr(), h.ws_post.numel()
    st.stream = stream_handle
    return tensors


def _fill_frcnn_step(st, h, stream_handle, rpn_logits, rpn_deltas, feat, cls_softmax, cls_deltas):
    """Fills the odet_fpn_step_t `st` as a SINGLE-LEVEL step (odet_fpn_step_t.single_level) of FrcnnHotPath slot `h`:
    rpn_logits [fh*fw, 2A] ([A bg | A fg]), rpn_deltas [fh*fw*A, 4], feat NHWC [1, fh, fw, C] in the slot's feature
    dtype.  Returns the tensors the caller must keep alive."""
    if isinstance(feat, (list, tuple)):
        feat = feat[0]
    for t in (rpn_logits, rpn_deltas, cls_softmax, cls_deltas):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise ValueError('FrcnnStepBatch.bind needs float32 contiguous GPU tensors')
    fdt = h.roi_features.dtype
    if not (feat.is_cuda and feat.dtype == fdt and feat.is_contiguous()):
        raise ValueError('FrcnnStepBatch.bind needs a %s contiguous GPU feature map' % fdt)
    if feat.dim() != 4 or feat.shape[0] != 1 or feat.shape[1] != h.fh or feat.shape[2] != h.fw or feat.shape[3] != h.C:
        raise ValueError('feature map must be NHWC [1,%d,%d,%d], got %s' % (h.fh, h.fw, h.C, tuple(feat.shape)))
    if rpn_logits.numel() != h.N * 2 or rpn_deltas.numel() != h.N * 4:
        raise ValueError('%d anchors expected, got rpn scores %s / deltas %s'
                         % (h.N, tuple(rpn_logits.shape), tuple(rpn_deltas.shape)))
    if cls_softmax.dim() != 2 or cls_softmax.shape[0] != h.K or cls_deltas.numel() != cls_softmax.numel() * 4:
        raise ValueError('class scores must be [%d, Ccls] and deltas [%d, Ccls, 4]' % (h.K, h.K))
    c = h.cfg
    st.single_level = 1
    st.roi_pool_mode = ops.ROI_POOL_MAX2 if h.max_pooling_flag else ops.ROI_POOL_NONE
    st.image_h, st.image_w = h.image_shape
    st.num_levels, st.A = 1, h.A
    st.fh[0], st.fw[0], st.stride[0] = h.fh, h.fw, h.stride
    flat = h.anchor_base.reshape(-1)
    for i in range(flat.shape[0]):
        st.wh[i] = float(flat[i])
    for k in range(4):
        st.rpn_means[k], st.rpn_stds[k] = float(c['rpn_means'][k]), float(c['rpn_stds'][k])
        st.roi_means[k], st.roi_stds[k] = float(c['roi_means'][k]), float(c['roi_stds'][k])
    st.num_proposals, st.rpn_nms_iou = h.K, float(c['rpn_nms_iou'])
    st.min_level, st.max_level, st.blind_chunks = 0, 0, h.blind_chunks
    st.nms_first_chunk = 0
    st.num_maps, st.channels, st.pool_size = 1, h.C, h.P
    st.maps_f16 = 1 if fdt == torch.float16 else 0
    st.maps[0].data, st.maps[0].H, st.maps[0].W, st.maps[0].stride = feat.data_ptr(), h.fh, h.fw, float(h.stride)
    st.ccls, st.num_classes = cls_softmax.shape[1], h.num_classes
    st.max_per_class, st.max_per_image = c['max_per_class'], c['max_per_image']
    st.nms_iou, st.score_threshold, st.min_edge = float(c['nms_iou']), float(c['score_threshold']), float(h.stride)
    st.rpn_logits, st.rpn_deltas = rpn_logits.data_ptr(), rpn_deltas.data_ptr()
    st.cls_scores, st.cls_deltas = cls_softmax.data_ptr(), cls_deltas.data_ptr()
    st.rois, st.roi_idx, st.roi_count = h.rois.data_ptr(), h.roi_idx.data_ptr(), h.roi_count.data_ptr()
    st.nms_done, st.sorted_rois = h.nms_done.data_ptr(), h.rois.data_ptr()       # (no level sort: the same list)
    st.roi_level, st.roi_perm, st.level_counts = None, None, None
    st.roi_features = h.roi_features.data_ptr()
    st.roi_order = h.roi_order.data_ptr() if h.roi_order is not None else None
    st.det_boxes, st.det_labels = h.det_boxes.data_ptr(), h.det_labels.data_ptr()
    st.det_scores, st.det_count, st.record = h.det_scores.data_ptr(), h.det_count.data_ptr(), h.record.data_ptr()
    st.ws_rpn, st.ws_rpn_bytes = h.ws_rpn.data_ptr(), h.ws_rpn.numel()
    st.ws_rpn_clean = 1
    st.ws_post_clean = 1
    st.ws_post, st.ws_post_bytes = h.ws_post.data_ptr(), h.ws_post.numel()
    st.stream = stream_handle
    return [rpn_logits, rpn_deltas, cls_softmax, cls_deltas, feat]


class FpnStepBatch:
    """Images through the FPN hot path in the SAME kernel launches, 8 per launch sequence (odet_fpn_step_enqueue_batch), on the
    CURRENT stream, stage by stage -- for a model that runs its dense RoI head between the stages (the
    assembled detectors): no enqueue thread, capturable into a HIP graph.  The RoI features of the images are
    consecutive blocks of one buffer, so the head can take them as one [B*K, ...] batch.

        sb = FpnStepBatch(8, image_shape, ...)
        sb.bind(b, rpn_logits[b], rpn_deltas[b], maps_of_image_b, cls_softmax[b], cls_deltas[b])   # every image
        sb.enqueue(STAGE_PROPOSALS | STAGE_ROI, B);  head on sb.roi_features[:B];  sb.enqueue(STAGE_DETECT, B)"""

    STAGE_PROPOSALS, STAGE_ROI, STAGE_DETECT = 1, 2, 4
    _slot_class, _fill = FpnHotPath, staticmethod(_fill_step)

    def __init__(self, max_batch, image_shape, num_classes=21, num_proposals=1000, channels=256, **kw):
        import ctypes as C
        if not 1 <= int(max_batch) <= 64:
            raise ValueError('max_batch must be in 1..64')
        self.n = int(max_batch)
        self.slots = [self._slot_class(image_shape, num_classes, num_proposals, channels, **kw) for _ in range(self.n)]
        h0 = self.slots[0]
        self.roi_features = torch.zeros((self.n,) + tuple(h0.roi_features.shape), dtype=h0.roi_features.dtype,
                                        device=h0.device)
        # (the images' NMS-completeness flags side by side: ONE device -> host copy checks a whole pass)
        self.nms_done_all = torch.zeros(self.n, dtype=torch.int32, device=h0.device)
        for b, h in enumerate(self.slots):
            h.roi_features = self.roi_features[b]
            h.nms_done = self.nms_done_all[b:b + 1]
        self.steps = [ops.L.OdetFpnStep() for _ in range(self.n)]
        # a launch sequence carries up to 8 images (ODET_MAX_STEP_BATCH: the kernels' per-image pointer tables); a larger
        # batch goes out as consecutive sequences of 8
        self._chunks = []
        for s0 in range(0, self.n, 8):
            m = min(8, self.n - s0)
            self._chunks.append((s0, (C.c_void_p * m)(*[C.addressof(st) for st in self.steps[s0:s0 + m]])))
        self._keep = [None] * self.n
        self._lib = ops.L.lib()

    def bind(self, b, rpn_logits, rpn_deltas, p_list, cls_softmax, cls_deltas):
        self._keep[b] = self._fill(self.steps[b], self.slots[b], 0, rpn_logits, rpn_deltas, p_list, cls_softmax,
                                   cls_deltas)

    def rebind(self, b, rpn_logits, rpn_deltas, p_list, cls_softmax, cls_deltas):
        """After one bind(b, ...): only the input pointers change (same shapes, dtypes, contiguity -- not
        re-checked here); a handful of stores instead of the whole descriptor."""
        st = self.steps[b]
        st.rpn_logits, st.rpn_deltas = rpn_logits.data_ptr(), rpn_deltas.data_ptr()
        st.cls_scores, st.cls_deltas = cls_softmax.data_ptr(), cls_deltas.data_ptr()
        if not isinstance(p_list, (list, tuple)):
            p_list = [p_list]
        for l in range(st.num_maps):
            st.maps[l].data = p_list[l].data_ptr()
        self._keep[b] = (rpn_logits, rpn_deltas, cls_softmax, cls_deltas) + tuple(p_list[:st.num_maps])

    def enqueue(self, stages, count=None):
        """The given stages of images 0..count-1 on the current stream (one launch sequence for all of them)."""
        count = self.n if count is None else int(count)
        handle = torch.cuda.current_stream().cuda_stream
        for st in self.steps[:count]:
            st.stream = handle
        for s0, arr in self._chunks:
            if s0 < count:
                ops.L.check(self._lib.odet_fpn_step_enqueue_batch(arr, min(8, count - s0), int(stages)))


class FrcnnStepBatch(FpnStepBatch):
    """FpnStepBatch for the single-level Faster R-CNN hot path (BASELINE configs 1-2: VGG16 / ResNet-C4): up to 8
    images through odet_frcnn_proposals -> RoI crops -> post-ops in the SAME launches (odet_fpn_step_t.single_level).

        sb = FrcnnStepBatch(4, image_shape, num_classes, 300, 1024, max_pooling_flag=False)
        sb.bind(b, rpn_logits[b], rpn_deltas[b], feat_nhwc[b:b+1], cls_softmax[b], cls_deltas[b])"""
    _slot_class, _fill = FrcnnHotPath, staticmethod(_fill_frcnn_step)

    def __init__(self, max_batch, image_shape, num_classes=21, num_proposals=300, channels=1024, **kw):
        super().__init__(max_batch, image_shape, num_classes, num_proposals, channels, **kw)


class FpnStreamPool:
    """Throughput arrangement: `n_streams` independent FpnHotPath slots, each with its own HIP stream
    and persistent buffers, fed by the library's native executor (one host thread per stream; a HIP
    launch costs ~3 us of host time and an image is ~11 launches, so a single enqueuing thread would
    cap the rate).  Images are independent (the path has no cross-image state), so consecutive images
    simply go to consecutive slots and their kernels overlap on the GPU: single-workgroup stages of
    one image (NMS scan, merge) run beside the chip-wide stages of another (RoI crops).

        pool = FpnStreamPool(4, image_shape, ...)
        pool.bind(slot, rpn_logits, rpn_deltas, p_list, cls_softmax, cls_deltas)   # once per buffer set
        pool.submit(slot)            # enqueue one image (all three stages) -- returns immediately
        pool.wait()                  # every submitted image has been enqueued; then sync the streams

    The stages of a submitted image are enqueued by a worker thread, so work that must follow them on
    the same stream from Python (a torch op) has to be issued after wait().  A model that runs a dense
    RoI head between the stages uses FpnHotPath directly."""

    _slot_class, _fill = FpnHotPath, staticmethod(_fill_step)

    def __init__(self, n_streams, image_shape, num_classes=21, num_proposals=1000, channels=256, batch=1, **kw):
        import ctypes as C
        self.n_streams = int(n_streams)
        self.batch = int(batch)
        if not 1 <= self.batch <= 8:
            raise ValueError('batch must be in 1..8 (ODET_MAX_STEP_BATCH)')
        self.n = self.n_streams * self.batch            # slots; slot k belongs to group k // batch
        sw = kw.pop('single_worker', None)              # (None: the measured rule below; True / False: for experiments)
        self.slots = [self._slot_class(image_shape, num_classes, num_proposals, channels, **kw) for _ in range(self.n)]
        # Batched groups need ~1.4 launches per image, far below one thread's launch rate, and ONE enqueue thread
        # issuing the groups in turn measured 3-4 % faster than one thread per stream (less contention inside the
        # HIP runtime); single-image launches (batch < 4) keep a thread per stream for the launch rate.
        self._single_worker = bool(sw) if sw is not None else self.batch >= 4
        self._group_streams = [torch.cuda.Stream() for _ in range(self.n_streams)]
        self.streams = [self._group_streams[k // self.batch] for k in range(self.n)]
        # (the slots' NMS-completeness flags side by side: ONE device -> host copy checks every image in flight)
        self.nms_done_all = torch.zeros(self.n, dtype=torch.int32, device=self.slots[0].device)
        for k, h in enumerate(self.slots):
            h.nms_done = self.nms_done_all[k:k + 1]
        self.nms_reruns = 0
        self.steps = [ops.L.OdetFpnStep() for _ in range(self.n)]
        self._groups = []
        for g in range(self.n_streams):
            arr = (C.c_void_p * self.batch)(*[C.addressof(self.steps[g * self.batch + j]) for j in range(self.batch)])
            self._groups.append(arr)
        self._keep = [None] * self.n
        self._C = C
        self._lib = ops.L.lib()
        self._exec = self._lib.odet_exec_create(self.n_streams)
        if not self._exec:
            raise ops.L.OdetError('odet_exec_create failed: %s' % self._lib.odet_last_error().decode())
        self._rr = 0

    def bind(self, slot, rpn_logits, rpn_deltas, p_list, cls_softmax, cls_deltas):
        """Points slot `slot` at one image's inputs (float32 contiguous GPU tensors, kept alive here)."""
        self._keep[slot] = self._fill(self.steps[slot], self.slots[slot], self.streams[slot].cuda_stream, rpn_logits,
                                      rpn_deltas, p_list, cls_softmax, cls_deltas)

    def submit(self, slot=None, stages=7):
        """Enqueue one image on `slot` (round-robin when None) as a launch sequence of its own.
        Returns the slot used."""
        if slot is None:
            slot = self._rr
            self._rr = (self._rr + 1) % self.n
        # (one enqueue thread per stream group -- or the single one: a group's launches always come from one thread)
        worker = 0 if self._single_worker else slot // self.batch
        ops.L.check(self._lib.odet_exec_submit(self._exec, worker, self._C.byref(self.steps[slot]), int(stages)))
        return slot

    def submit_group(self, group=None, stages=7):
        """Enqueue the `batch` images of stream group `group` (round-robin when None) in the SAME kernel
        launches (odet_fpn_step_enqueue_batch).  Returns the group used."""
        if group is None:
            group = self._rr % self.n_streams
            self._rr = (self._rr + 1) % self.n_streams
        worker = 0 if self._single_worker else group
        ops.L.check(self._lib.odet_exec_submit_batch(self._exec, worker, self._groups[group], self.batch, int(stages)))
        return group

    def enqueue_group(self, group, stages=7):
        """The `batch` images of stream group `group` enqueued by the CALLING thread (odet_fpn_step_enqueue_batch on the
        group's stream; returns when the launches are in the stream) -- for loops that follow every group with work
        of their own in stream order (the multi-rank exchange of parallel.GroupExchange): no hand-off to the enqueue
        thread, hence no host wait before the follow-up can be issued.  A batched group is ~12 launches for 8 images
        (~40 us of host time against ~270 us of GPU time), so one thread keeps up.  Do not mix with submit_group() on
        the same group without a wait() in between (two threads would enqueue onto one stream)."""
        ops.L.check(self._lib.odet_fpn_step_enqueue_batch(self._groups[group], self.batch, int(stages)))
        return group

    def wait(self):
        rc = self._lib.odet_exec_wait(self._exec)
        if rc != 0:
            msg = self._lib.odet_exec_last_error(self._exec).decode()
            # a failed launch sequence may have left a post-ops ticket or an NMS header half-way: the "clean workspace" promises
            # of the step descriptors (ws_rpn_clean / ws_post_clean) hold again only after the workspaces are zero-filled
            try:
                torch.cuda.synchronize()
                for h in self.slots:
                    h.ws_post.zero_()
                    h.ws_rpn.zero_()
                torch.cuda.synchronize()
            except Exception:
                pass
            raise ops.L.OdetError('odet error %d: %s' % (rc, msg))

    def recover_incomplete(self):
        """After wait(): the slots whose sync-free NMS did not complete inside their chunks (nms_done = 0: reported EMPTY)
        go through the hot path again in the EXACT mode (nms_done = NULL: the library checks every chunk on the host and
        runs as many as the image needs -- model/region_proposal.py:73-81 is always exact), each as a launch sequence of its
        own on its stream, enqueued by the calling thread.  Returns the slots re-run (also counted in `nms_reruns`); their
        outputs and records are complete once their streams have run (the exact mode synchronises them itself)."""
        for st in self._group_streams:
            st.synchronize()
        bad = [k for k, f in enumerate(self.nms_done_all.tolist()) if f != 1]
        for k in bad:
            st = self.steps[k]
            keep = st.nms_done
            st.nms_done, st.ws_rpn_clean = None, 0     # (exact mode; the library zeroes the workspace header itself)
            try:
                ops.L.check(self._lib.odet_fpn_step_enqueue(self._C.byref(st), 7))
            finally:
                st.nms_done, st.ws_rpn_clean = keep, 1
            with torch.cuda.stream(self.streams[k]):
                self.nms_done_all[k:k + 1].fill_(1)
        for k in bad:
            self.streams[k].synchronize()
        self.nms_reruns += len(bad)
        return bad

    def close(self):
        if self._exec:
            self._lib.odet_exec_wait(self._exec)
            self._lib.odet_exec_destroy(self._exec)
            self._exec = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class FrcnnStreamPool(FpnStreamPool):
    """FpnStreamPool for the single-level Faster R-CNN hot path: stream groups of `batch` images that share their
    launches, fed by the native executor -- configs 1-2 shard over GPUs exactly like the FPN path (parallel.py)."""
    _slot_class, _fill = FrcnnHotPath, staticmethod(_fill_frcnn_step)

    def __init__(self, n_streams, image_shape, num_classes=21, num_proposals=300, channels=1024, batch=1, **kw):
        super().__init__(n_streams, image_shape, num_classes, num_proposals, channels, batch=batch, **kw)


def synthetic_frcnn_inputs(image_shape, num_classes=21, num_proposals=300, channels=1024, stride=16, A=9, seed=1234,
                           device='cuda'):
    """Seeded synthetic inputs of one image for the single-level hot path (SURVEY 8d recipe), numpy + GPU tensors."""
    import math
    rng = np.random.default_rng(seed)
    fh, fw = int(math.ceil(image_shape[0] / stride)), int(math.ceil(image_shape[1] / stride))
    n = fh * fw * A
    host = dict(feat=rng.standard_normal((1, fh, fw, channels), dtype=np.float32),
                rpn_logits=rng.normal(0, 2.0, (fh * fw, 2 * A)).astype(np.float32), rpn_deltas=syn.rpn_deltas(n, rng, 0.1),
                cls_scores=syn.class_scores(num_proposals, num_classes, rng),
                cls_deltas=syn.class_deltas(num_proposals, num_classes, rng))
    dev = {k: torch.from_numpy(v).to(device) for k, v in host.items()}
    return host, dev


def synthetic_fpn_inputs(image_shape, num_classes=21, num_proposals=1000, channels=256, seed=1234,
                         score_kind='distinct', device='cuda'):
    """Seeded synthetic inputs of SURVEY.md section 8(d) for one image, as numpy + GPU tensors."""
    rng = np.random.default_rng(seed)
    shapes = syn.fpn_level_shapes(image_shape)
    n = syn.num_fpn_anchors(image_shape)
    feats = syn.features(shapes[:4], channels, rng)
    deltas = syn.rpn_deltas(n, rng, 0.1)
    if score_kind == 'distinct':
        prob = syn.scores_distinct(n, rng)
    else:
        from .utils.anchor_generator import _wh_table  # anchors on the host only to place the clusters
        anchors = _host_fpn_anchors(image_shape)
        prob = syn.scores_clustered(anchors, image_shape, rng)
    logits = syn.logits_from_prob(prob, rng)
    cls_scores = syn.class_scores(num_proposals, num_classes, rng)
    cls_deltas = syn.class_deltas(num_proposals, num_classes, rng)
    host = dict(feats=feats, rpn_deltas=deltas, rpn_logits=logits, cls_scores=cls_scores, cls_deltas=cls_deltas)
    dev = {k: ([torch.from_numpy(x).to(device) for x in v] if isinstance(v, list) else torch.from_numpy(v).to(device))
           for k, v in host.items()}
    return host, dev


def _host_fpn_anchors(image_shape):
    """numpy FPN anchors (only used to build clustered synthetic scores)."""
    from .utils.anchor_generator import _wh_table
    out = []
    for (fh, fw), s, b in zip(syn.fpn_level_shapes(image_shape), syn.FPN_STRIDES, syn.FPN_BASE_SIZES):
        wh = _wh_table(b, syn.FPN_SCALES, syn.FPN_RATIOS)
        xs = np.arange(fw, dtype=np.float32) * np.float32(s)
        ys = np.arange(fh, dtype=np.float32) * np.float32(s)
        cx = np.tile(xs, fh)[:, None]
        cy = np.repeat(ys, fw)[:, None]
        hw = np.float32(0.5) * wh[None, :, 0]
        hh = np.float32(0.5) * wh[None, :, 1]
        out.append(np.stack([cx - hw, cy - hh, cx + hw, cy + hh], axis=2).reshape(-1, 4))
    return np.concatenate(out, axis=0).astype(np.float32)

"""ctypes binding of libodet_hip.so (include/odet.h).

PyTorch is only plumbing here: it owns device memory and the HIP stream; every detection op
is a hand-written HIP kernel behind the C ABI.  There is NO CPU fallback: if the library is
missing, or a tensor is not on the GPU, the call raises.
"""
import ctypes as C
import os

import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, 'libodet_hip.so')     # (tools/_diag.py points it at a diagnostic build; no environment switch)
ODET_VERSION = 103                                   # include/odet.h

_lib = None

_vp, _i, _f, _i64, _sz = C.c_void_p, C.c_int, C.c_float, C.c_int64, C.c_size_t


class OdetLevel(C.Structure):
    """odet_level_t"""
    _fields_ = [('data', C.c_void_p), ('H', C.c_int32), ('W', C.c_int32), ('stride', C.c_float)]


class OdetConvLevel(C.Structure):
    """odet_conv_level_t"""
    _fields_ = [('x', C.c_void_p), ('y', C.c_void_p), ('H', C.c_int32), ('W', C.c_int32)]


MAX_LEVELS = 8
MAX_ANCHORS_PER_CELL = 32


class OdetFpnStep(C.Structure):
    """odet_fpn_step_t (include/odet.h); size checked against odet_fpn_step_sizeof() at load time."""
    _fields_ = [
        ('image_h', C.c_int32), ('image_w', C.c_int32), ('num_levels', C.c_int32), ('A', C.c_int32),
        ('fh', C.c_int32 * MAX_LEVELS), ('fw', C.c_int32 * MAX_LEVELS), ('stride', C.c_int32 * MAX_LEVELS),
        ('wh', C.c_float * (MAX_LEVELS * MAX_ANCHORS_PER_CELL * 2)),
        ('rpn_means', C.c_float * 4), ('rpn_stds', C.c_float * 4),
        ('num_proposals', C.c_int32), ('rpn_nms_iou', C.c_float),
        ('min_level', C.c_int32), ('max_level', C.c_int32), ('blind_chunks', C.c_int32), ('nms_first_chunk', C.c_int32),
        ('num_maps', C.c_int32), ('channels', C.c_int32), ('pool_size', C.c_int32), ('maps_f16', C.c_int32),
        ('maps', OdetLevel * MAX_LEVELS),
        ('ccls', C.c_int32), ('num_classes', C.c_int32), ('max_per_class', C.c_int32), ('max_per_image', C.c_int32),
        ('roi_means', C.c_float * 4), ('roi_stds', C.c_float * 4),
        ('nms_iou', C.c_float), ('score_threshold', C.c_float), ('min_edge', C.c_float),
        ('rpn_logits', C.c_void_p), ('rpn_deltas', C.c_void_p), ('cls_scores', C.c_void_p), ('cls_deltas', C.c_void_p),
        ('rois', C.c_void_p), ('roi_idx', C.c_void_p), ('roi_count', C.c_void_p), ('nms_done', C.c_void_p),
        ('sorted_rois', C.c_void_p), ('roi_level', C.c_void_p), ('roi_perm', C.c_void_p),
        ('level_counts', C.c_void_p), ('roi_features', C.c_void_p), ('roi_order', C.c_void_p),
        ('det_boxes', C.c_void_p), ('det_labels', C.c_void_p), ('det_scores', C.c_void_p),
        ('det_count', C.c_void_p), ('record', C.c_void_p),
        ('ws_rpn', C.c_void_p), ('ws_rpn_bytes', C.c_size_t),
        ('ws_post', C.c_void_p), ('ws_post_bytes', C.c_size_t),
        ('stream', C.c_void_p),
        ('roi_start_event', C.c_void_p), ('roi_stop_event', C.c_void_p),
        ('ws_rpn_clean', C.c_int32), ('single_level', C.c_int32), ('roi_pool_mode', C.c_int32),
        ('ws_post_clean', C.c_int32),
    ]


# name -> (restype, argtypes); mirrors include/odet.h one to one
SIGNATURES = {
    'odet_version': (_i, []),
    'odet_last_error': (C.c_char_p, []),
    'odet_anchors_shift': (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    'odet_anchors_fpn': (_i, [_i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    'odet_decode': (_i, [_vp, _vp, _i64, _i, _vp, _vp, _i, _i, _vp, _vp]),
    'odet_encode': (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp]),
    'odet_clip': (_i, [_vp, _i, _f, _i, _i, _vp, _vp]),
    'odet_compact_workspace_bytes': (_sz, [_i]),
    'odet_clip_filter': (_i, [_vp, _i, _f, _i, _i, _f, _vp, _vp, _vp, _vp, _sz, _vp]),
    'odet_range_filter': (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    'odet_where_greater': (_i, [_vp, _i64, _i, _f, _vp, _vp, _vp, _sz, _vp]),
    'odet_pairwise_iou': (_i, [_vp, _i, _vp, _i, _vp, _vp]),
    'odet_gather_rows': (_i, [_vp, _vp, _i, _i, _vp, _i, _vp, _vp]),
    'odet_rpn_fg_softmax': (_i, [_vp, _i, _i, _i, _vp, _vp]),
    'odet_nms_workspace_bytes': (_sz, [_i, _i]),
    'odet_nms': (_i, [_vp, _vp, _i, _i, _f, _vp, _vp, _vp, _i, _vp, _vp, _sz, _vp]),
    'odet_region_proposal_workspace_bytes': (_sz, [_i, _i]),
    'odet_region_proposal': (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _i, _f, _vp, _vp, _vp, _i, _vp, _vp, _sz,
                                   _vp]),
    'odet_fpn_proposals_workspace_bytes': (_sz, [_i, _i]),
    'odet_fpn_proposals': (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _i, _f, _i, _i,
                                 _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _sz, _vp]),
    'odet_frcnn_proposals_workspace_bytes': (_sz, [_i, _i]),
    'odet_frcnn_proposals': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _i, _f, _vp, _vp, _vp, _i, _vp,
                                   _vp, _sz, _vp]),
    'odet_assign_levels': (_i, [_vp, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    'odet_roi_pool': (_i, [_vp, _i, _i, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    'odet_roi_pool_timed': (_i, [_vp, _i, _i, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    'odet_roi_order': (_i, [_vp, _vp, _i, _vp, _i, _i, _vp, _vp]),
    'odet_roi_pool_ordered': (_i, [_vp, _i, _i, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    'odet_roi_pool_f16': (_i, [_vp, _i, _i, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    'odet_roi_pool_f16_timed': (_i, [_vp, _i, _i, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    'odet_prof_event_create': (_i, [_vp]),
    'odet_prof_event_destroy': (_i, [_vp]),
    'odet_prof_event_elapsed_ms': (_i, [_vp, _vp, _vp]),
    'odet_post_ops_workspace_bytes': (_sz, [_i, _i]),
    'odet_post_ops': (_i, [_vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _vp, _vp, _i, _i, _f, _f, _f,
                           _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    'odet_post_ops_record': (_i, [_vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _vp, _vp, _i, _i, _f, _f, _f,
                                  _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    'odet_eval_detect': (_i, [_vp, _vp, _vp, _i, _vp, _i, _i, _f, _f, _f, _vp, _vp, _i, _i, _f, _f, _f,
                              _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    'odet_pack_detections': (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    'odet_fpn_step_sizeof': (_sz, []),
    'odet_fpn_topdown_merge': (_i, [_vp, _i, _i, _vp, _i, _i, _i, _i, _vp, _i, _vp]),
    'odet_bias_act': (_i, [_vp, _vp, _vp, C.c_longlong, _i, _i, _i, _vp]),
    'odet_rpn_pack': (_i, [_vp, _vp, C.c_longlong, _i, _i, _vp, C.c_longlong, C.c_longlong, _i, _vp]),
    'odet_bias_relu_maxpool': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'odet_rpn_head_tail_f16': (_i, [_vp, _vp, _vp, _vp, C.c_longlong, _i, _i, _vp, C.c_longlong, C.c_longlong, _vp,
                                    C.c_longlong, C.c_longlong, _vp]),
    'odet_conv3x3_f16': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'odet_conv3x3_relu_pool2_f16': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'odet_calib_read_rows': (_i, [_vp, C.c_ulonglong, _i, _vp, _vp]),
    'odet_calib_stream_mix': (_i, [_vp, C.c_ulonglong, _vp, C.c_ulonglong, _vp, _vp, _vp]),
    'odet_rpn_head_fused_f16': (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, C.c_longlong, _vp, C.c_longlong, _vp]),
    'odet_bottleneck_tail_f16': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'odet_conv3x3_conv1x1_f16': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'odet_stem_pack_weights_f16': (_i, [_vp, C.c_longlong, C.c_longlong, C.c_longlong, C.c_longlong, _vp, _vp]),
    'odet_stem_conv7_pool3_f16': (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _vp]),
    'odet_conv3x3_rgb_pack_weights_f16': (_i, [_vp, C.c_longlong, C.c_longlong, C.c_longlong, C.c_longlong, _vp, _vp]),
    'odet_conv3x3_rgb_f16': (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'odet_conv3x3_f32': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'odet_conv3x3_f32_levels': (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _i, _vp]),
    'odet_conv3x3_f16_levels': (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _i, _vp]),
    'odet_pointwise_f16': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'odet_pointwise_f32': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'odet_lateral_merge_f32': (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _i, _i, _i, _i, _i, _vp]),
    'odet_pointwise_dual_f32': (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _i, _vp]),
    'odet_x3_workspace_bytes': (_sz, []),
    'odet_x2_status_offset': (_sz, []),
    'odet_split_bf16x3': (_i, [_vp, _vp, _i64, _vp]),
    'odet_conv3x3_x3': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    'odet_conv3x3_x3_levels': (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _i, _vp, _sz, _vp]),
    'odet_pointwise_x3': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    'odet_lateral_merge_x3': (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    'odet_pointwise_dual_x3': (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _i, _vp, _sz, _vp]),
    'odet_split_f16x2': (_i, [_vp, _vp, _i64, _i, _vp]),
    'odet_conv3x3_x2': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    'odet_conv3x3_x2_levels': (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    'odet_pointwise_x2': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    'odet_lateral_merge_x2': (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    'odet_pointwise_dual_x2': (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _sz, _vp]),
    'odet_stem_patches_f32': (_i, [_vp, _vp, _i, _i, _i, _vp]),
    'odet_rgb_patches3x3_f32': (_i, [_vp, _vp, _i, _i, _i, _vp]),
    'odet_pointwise_dual_f16': (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _i, _vp]),
    'odet_dense_f16_out_f32': (_i, [_vp, _vp, _vp, _vp, C.c_longlong, _i, _i, _i, _vp]),
    'odet_lateral_merge_f16': (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _i, _i, _i, _i, _i, _vp]),
    'odet_conv1x1_f16': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, C.c_longlong, _i, _i, _i, _vp]),
    'odet_rpn_pack_pair': (_i, [_vp, _vp, C.c_longlong, _i, _i, _vp, C.c_longlong, C.c_longlong, _vp, C.c_longlong,
                                C.c_longlong, _i, _vp]),
    'odet_fpn_step_enqueue': (_i, [_vp, _i]),
    'odet_fpn_step_enqueue_batch': (_i, [_vp, _i, _i]),
    'odet_exec_create': (_vp, [_i]),
    'odet_exec_destroy': (None, [_vp]),
    'odet_exec_submit': (_i, [_vp, _i, _vp, _i]),
    'odet_exec_submit_batch': (_i, [_vp, _i, _vp, _i, _i]),
    'odet_exec_wait': (_i, [_vp]),
    'odet_exec_last_error': (C.c_char_p, [_vp]),
}


class OdetError(RuntimeError):
    pass


def lib():
    """Loads libodet_hip.so once.  Raises if it has not been built -- never falls back."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OdetError(
                'libodet_hip.so is missing (%s). Build it with `python -c "import __graft_entry__ as g; '
                'g.build()"` or `python -m tf_eager_object_detection_amd._build`; there is no CPU fallback.'
                % LIB_PATH)
        # torch is imported first on purpose: its bundled libamdhip64.so (SONAME libamdhip64.so.7)
        # is then the one HIP runtime of the process and this library binds to it.
        handle = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        if handle.odet_version() != ODET_VERSION:
            raise OdetError('libodet_hip.so version mismatch: %d' % handle.odet_version())
        if handle.odet_fpn_step_sizeof() != C.sizeof(OdetFpnStep):
            raise OdetError('odet_fpn_step_t layout mismatch: library %d bytes, binding %d'
                            % (handle.odet_fpn_step_sizeof(), C.sizeof(OdetFpnStep)))
        _lib = handle
    return _lib


_recording = None


class recording:
    """Context manager: collects the (function, ctypes args) of every C-ABI call made through call()
    so that a caller with persistent buffers (pipeline.FpnHotPath) can replay them without
    re-marshalling the arguments -- a host-side launch plan; the library itself is stateless."""

    def __enter__(self):
        global _recording
        self.calls = []
        self._prev = _recording
        _recording = self.calls
        return self.calls

    def __exit__(self, *exc):
        global _recording
        _recording = self._prev
        return False


def call(name, *args):
    fn = getattr(lib(), name)
    if _recording is not None:
        _recording.append((fn, args))
    rc = fn(*args)
    if rc != 0:
        check(rc)


def check(rc):
    if rc != 0:
        raise OdetError('odet error %d: %s' % (rc, lib().odet_last_error().decode('utf-8', 'replace')))


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def dptr(t, dtype=None, name='tensor'):
    """Device pointer of a contiguous CUDA/HIP tensor (None -> NULL)."""
    if t is None:
        return None
    if not isinstance(t, torch.Tensor):
        raise TypeError('%s must be a torch.Tensor, got %s' % (name, type(t).__name__))
    if not t.is_cuda:
        raise OdetError('%s must live on the GPU: tf_eager_object_detection_amd has no CPU path' % name)
    if dtype is not None and t.dtype != dtype:
        raise TypeError('%s must be %s, got %s' % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError('%s must be contiguous' % name)
    return C.c_void_p(t.data_ptr())


def f32c(t, name='tensor'):
    """float32 contiguous view/copy of a GPU tensor (the reference casts with tf.to_float)."""
    if not isinstance(t, torch.Tensor):
        raise TypeError('%s must be a torch.Tensor, got %s' % (name, type(t).__name__))
    if not t.is_cuda:
        raise OdetError('%s must live on the GPU: tf_eager_object_detection_amd has no CPU path' % name)
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def host4(values, name):
    """Python sequence of 4 floats -> C float[4] (means / stds are host data in the reference)."""
    if values is None:
        raise ValueError('%s is None' % name)
    vals = [float(v) for v in values]
    if len(vals) != 4:
        raise ValueError('%s must have 4 elements' % name)
    return (C.c_float * 4)(*vals)


def workspace(nbytes, device):
    return torch.empty(int(nbytes), dtype=torch.uint8, device=device)

"""RoI feature extraction -- counterpart of the reference's object_detection/model/roi_pooling.py.

All three reference layers map onto ONE fused HIP kernel (odet_roi_pool): TF crop_and_resize
sampling + the 2x2 max / avg pool, without materialising the [R,14,14,C] crops."""
import torch

from .. import ops

__all__ = ['RoiPoolingCropAndResize', 'RoiPoolingRoiAlign', 'RoiPoolingCropAndResize2', 'crop_and_resize',
           'roi_align', 'roi_pooling_fpn_levels']


def _spatial_order(feature_map, rois, stride):
    """processing order of a single-level layer's RoIs (odet_roi_order: sorted by y, x centre) -- consecutive workgroups, which
    share an XCD, then tap one neighbourhood of the map, so every XCD's L2 fetches its part of the map once instead of most of
    it (round 4: the tensorpack RoIAlign layer moved 1.84 x its distinct bytes without it).  Results are the same bits in any
    order.  None for launches too small to pay for the extra launch."""
    n = int(rois.shape[0])
    if n < 128 or n > 8192:
        return None
    h, w = int(feature_map.shape[1]), int(feature_map.shape[2])
    return ops.roi_order(rois, None, (max(1, int(round(h * float(stride)))), max(1, int(round(w * float(stride))))))


class RoiPoolingCropAndResize2(torch.nn.Module):
    """reference model/roi_pooling.py:8-42 (FPN variant: boxes normalised by the IMAGE size,
    always crop 2P + 2x2 max-pool)."""

    def __init__(self, pool_size):
        super().__init__()
        self._pool_size = pool_size

    @torch.no_grad()
    def forward(self, inputs, training=None, mask=None):
        shared_layers, rois, image_shape = inputs
        return ops.roi_pool([shared_layers], rois, None, ops.ROI_NORM_IMAGE, self._pool_size, ops.ROI_POOL_MAX2,
                            image_shape=image_shape)

    call = forward


class RoiPoolingCropAndResize(torch.nn.Module):
    """reference model/roi_pooling.py:45-90 (boxes / stride / (dim-1); crop 2P + max-pool when
    ``max_pooling_flag`` else crop P)."""

    def __init__(self, pool_size, max_pooling_flag=True):
        super().__init__()
        self._pool_size = pool_size
        self._max_pooling_flag = max_pooling_flag

    @torch.no_grad()
    def forward(self, inputs, training=None, mask=None):
        shared_layers, rois, extractor_stride = inputs
        mode = ops.ROI_POOL_MAX2 if self._max_pooling_flag else ops.ROI_POOL_NONE
        return ops.roi_pool([shared_layers], rois, None, ops.ROI_NORM_STRIDE, self._pool_size, mode,
                            strides=[float(extractor_stride)], order=_spatial_order(shared_layers, rois, extractor_stride))

    call = forward


def crop_and_resize(image, boxes, box_ind, crop_size, pad_border=True):
    """reference model/roi_pooling.py:93-137 (tensorpack-style aligned crop).  ``boxes`` are in
    feature-map coordinates; ``box_ind`` must be all zeros (batch = 1, as everywhere in the
    reference: roi_pooling.py:28,66,152)."""
    assert isinstance(crop_size, int), crop_size
    mode = ops.ROI_NORM_TP_ALIGN if pad_border else 3
    return ops.roi_pool([image], boxes, None, mode, crop_size, ops.ROI_POOL_NONE, strides=[1.0],
                        order=_spatial_order(image, boxes, 1.0))


def roi_align(featuremap, boxes, resolution):
    """reference model/roi_pooling.py:140-155: 4 samples per bin (crop 2*resolution) + 2x2 avg."""
    return ops.roi_pool([featuremap], boxes, None, ops.ROI_NORM_TP_ALIGN, resolution, ops.ROI_POOL_AVG2,
                        strides=[1.0], order=_spatial_order(featuremap, boxes, 1.0))


class RoiPoolingRoiAlign(torch.nn.Module):
    """reference model/roi_pooling.py:158-177."""

    def __init__(self, pool_size):
        super().__init__()
        self._pool_size = pool_size

    @torch.no_grad()
    def forward(self, inputs, training=None, mask=None):
        shared_layers, rois, extractor_stride = inputs
        return ops.roi_pool([shared_layers], rois, None, ops.ROI_NORM_TP_ALIGN, self._pool_size,
                            ops.ROI_POOL_AVG2, strides=[float(extractor_stride)],
                            order=_spatial_order(shared_layers, rois, extractor_stride))

    call = forward


def roi_pooling_fpn_levels(p_list, rois_sorted, roi_level, image_shape, pool_size, count_dev=None, out=None):
    """Native addition: what reference model/fpn/base_fpn_model.py:152-161 (_get_roi_features)
    does with one RoiPoolingCropAndResize2 call per non-empty level + concat, as ONE launch over
    the level-sorted RoIs (``roi_level`` = 0-based level of each row)."""
    return ops.roi_pool(list(p_list), rois_sorted, roi_level, ops.ROI_NORM_IMAGE, pool_size, ops.ROI_POOL_MAX2,
                        image_shape=image_shape, count_dev=count_dev, out=out)

"""BaseFPN / RpnHead / ResnetV1Fpn -- counterparts of the reference's model/fpn/base_fpn_model.py (BaseFPN :14-390, RpnHead
:393-434) and model/fpn/resnet_fpn.py (ResnetV1Fpn :410-543): the CALLER objects of the hot path, with the reference's
constructor arguments, `call(inputs, training=None, mask=None)` and `im_detect(preprocessed_img, img_scale)`.

`call` composes the reference-NAMED layers of this package in the reference's order -- `_extractor` -> `_neck` ->
`_rpn_head` per level -> `_get_anchors` (make_anchors per level) -> fg softmax -> `_rpn_proposal` (RegionProposal) ->
`_assign_levels` -> `_get_roi_features` (one RoiPoolingCropAndResize2 call per non-empty level + concat) -> `_roi_head` ->
softmax -> post_ops_prediction -- with the reference's dynamic shapes (one host sync per stage, as TF-eager has) and its
batch of one image (model/roi_pooling.py:28).  It is the drop-in reading of "base_fpn_model.py can call it"; the
static-shape, sync-free arrangement of the same stages for throughput is model/fpn_detector.py + pipeline.FpnHotPath.
Every tensor op between the layers that the reference delegates to TF (softmax, gather, concat, reshape) is a torch op on
the GPU; every detection op is a HIP kernel behind the C ABI.  Training (`training=True`) returns the four losses of the
forward pass (:232-264) on the target layers of this package; the dense kernels are inference kernels (no backward)."""
import torch

from .. import ops
from ..utils.anchor_generator import make_anchors
from .anchor_target import AnchorTarget
from .fpn_detector import caller_range_checked
from .losses import cls_loss, smooth_l1_loss
from .prediction import post_ops_prediction
from .proposal_target import ProposalTarget
from .region_proposal import RegionProposal
from .roi_pooling import RoiPoolingCropAndResize2

__all__ = ['BaseFPN', 'RpnHead', 'ResnetV1Fpn']


def _image_nhwc(image):
    """the reference's input: one preprocessed image [1,H,W,3] float (model/roi_pooling.py:28: batch 1)"""
    if not isinstance(image, torch.Tensor) or image.dim() != 4 or image.shape[0] != 1 or image.shape[3] != 3:
        raise ValueError('image must be a [1,H,W,3] tensor (the reference runs one image per call)')
    if not image.is_cuda:
        raise ops.L.OdetError('image must live on the GPU: tf_eager_object_detection_amd has no CPU path')
    return image.float().contiguous()


class RpnHead(torch.nn.Module):
    """reference base_fpn_model.py:393-434: 3x3 conv 512 + ReLU -> 1x1 score conv (2A) / 1x1 bbox conv (4A), outputs reshaped
    to [-1, 2] / [-1, 4].  The layers live in `dense` (a detector of model/fpn_detector.py: rpn_conv, rpn_score, rpn_bbox); the
    whole head of a level is one launch in float16 (ops.rpn_head_fused) and three in float32."""

    def __init__(self, num_anchors, weight_decay=0.0001, dense=None):
        super().__init__()
        self._num_anchors = num_anchors
        self._dense = [dense]                    # (not a sub-module: the parameters belong to the model that owns `dense`)
        if dense is not None and dense.A != num_anchors:
            raise ValueError('RpnHead: %d anchors per cell but the dense part was built for %d' % (num_anchors, dense.A))

    @torch.no_grad()
    def forward(self, inputs, training=None, mask=None):
        scores, deltas = self._dense[0].rpn([inputs])
        return scores.reshape(-1, 2), deltas.reshape(-1, 4)

    call = forward


class BaseFPN(torch.nn.Module):
    """reference base_fpn_model.py:14-141 (same constructor arguments, same private attribute names)."""

    def __init__(self, roi_feature_size=(7, 7, 256), num_classes=21, weight_decay=0.0001,
                 level_name_list=('p2', 'p3', 'p4', 'p5', 'p6'), min_level=2, max_level=5,
                 anchor_stride_list=(4, 8, 16, 32, 64), base_anchor_size_list=(32, 64, 128, 256, 512),
                 ratios=(0.5, 1.0, 2.0), scales=(1.,),
                 rpn_proposal_means=(0, 0, 0, 0), rpn_proposal_stds=(1.0, 1.0, 1.0, 1.0),
                 rpn_proposal_num_pre_nms_train=12000, rpn_proposal_num_post_nms_train=2000,
                 rpn_proposal_num_pre_nms_test=6000, rpn_proposal_num_post_nms_test=300,
                 rpn_proposal_nms_iou_threshold=0.7,
                 rpn_sigma=3.0, rpn_training_pos_iou_threshold=0.7, rpn_training_neg_iou_threshold=0.3,
                 rpn_training_total_num_samples=256, rpn_training_max_pos_samples=128,
                 roi_proposal_means=(0, 0, 0, 0), roi_proposal_stds=(0.1, 0.1, 0.2, 0.2),
                 roi_pool_size=7, roi_pooling_max_pooling_flag=True,
                 roi_sigma=1, roi_training_pos_iou_threshold=0.5, roi_training_neg_iou_threshold=0.1,
                 roi_training_total_num_samples=128, roi_training_max_pos_samples=32,
                 prediction_max_objects_per_image=50, prediction_max_objects_per_class=50,
                 prediction_nms_iou_threshold=0.3, prediction_score_threshold=0.):
        super().__init__()
        self.roi_feature_size = roi_feature_size
        self.num_classes = num_classes
        self.weight_decay = weight_decay
        self._level_name_list = level_name_list
        self._min_level = min_level
        self._max_level = max_level
        self._anchor_stride_list = anchor_stride_list
        self._base_anchor_size_list = base_anchor_size_list
        self._ratios = ratios
        self._scales = scales
        self._num_anchors = len(ratios) * len(scales)
        self._rpn_sigma = rpn_sigma
        self._roi_sigma = roi_sigma
        self._roi_proposal_means = roi_proposal_means
        self._roi_proposal_stds = roi_proposal_stds
        self._prediction_max_objects_per_image = prediction_max_objects_per_image
        self._prediction_max_objects_per_class = prediction_max_objects_per_class
        self._prediction_nms_iou_threshold = prediction_nms_iou_threshold
        self._prediction_score_threshold = prediction_score_threshold

        self._extractor = self._get_extractor()
        self._neck = self._get_neck()
        self._rpn_head = self._get_rpn_head(weight_decay)
        self._rpn_proposal = RegionProposal(
            num_anchors=self._num_anchors, num_pre_nms_train=rpn_proposal_num_pre_nms_train,
            num_post_nms_train=rpn_proposal_num_post_nms_train, num_pre_nms_test=rpn_proposal_num_pre_nms_test,
            num_post_nms_test=rpn_proposal_num_post_nms_test, nms_iou_threshold=rpn_proposal_nms_iou_threshold,
            target_means=rpn_proposal_means, target_stds=rpn_proposal_stds)
        self._roi_pooling = RoiPoolingCropAndResize2(pool_size=roi_pool_size)
        self._roi_head = self._get_roi_head()
        self._anchor_target = AnchorTarget(
            pos_iou_threshold=rpn_training_pos_iou_threshold, neg_iou_threshold=rpn_training_neg_iou_threshold,
            total_num_samples=rpn_training_total_num_samples, max_pos_samples=rpn_training_max_pos_samples,
            target_means=rpn_proposal_means, target_stds=rpn_proposal_stds)
        self._proposal_target = ProposalTarget(
            num_classes=num_classes, pos_iou_threshold=roi_training_pos_iou_threshold,
            neg_iou_threshold=roi_training_neg_iou_threshold, total_num_samples=roi_training_total_num_samples,
            max_pos_samples=roi_training_max_pos_samples, target_means=roi_proposal_means,
            target_stds=roi_proposal_stds)

    def _get_roi_head(self):
        raise NotImplementedError

    def _get_extractor(self):
        raise NotImplementedError

    def _get_neck(self):
        raise NotImplementedError

    def _get_rpn_head(self, weight_decay):
        return RpnHead(num_anchors=self._num_anchors, weight_decay=weight_decay)

    # ---- :143-200 ------------------------------------------------------------------------------
    def _get_roi_features(self, rois_list, p_list, image_shape):
        all_roi_features = []
        for level_name, cur_rois, cur_p, cur_stride in zip(self._level_name_list[:-1], rois_list, p_list,
                                                           self._anchor_stride_list):
            if cur_rois.shape[0] == 0:
                continue
            all_roi_features.append(self._roi_pooling((cur_p, cur_rois, image_shape)))
        return torch.cat(all_roi_features, dim=0)

    def _get_anchors(self, image_shape):
        all_anchors = []
        for idx in range(len(self._level_name_list)):
            extractor_stride = self._anchor_stride_list[idx]
            # (python's true division + ceil, as tf.ceil(image_shape[0] / extractor_stride) on python ints)
            all_anchors.append(make_anchors(base_anchor_size=self._base_anchor_size_list[idx], anchor_scales=self._scales,
                                            anchor_ratios=self._ratios,
                                            featuremap_height=float(-(-image_shape[0] // extractor_stride)),
                                            featuremap_width=float(-(-image_shape[1] // extractor_stride)),
                                            stride=extractor_stride))
        return torch.cat(all_anchors, dim=0)

    def _get_fpn_head_results(self, p_list):
        all_fpn_scores, all_fpn_bbox_pred = [], []
        for level_name, p in zip(self._level_name_list, p_list):
            cur_score, cur_bboxes_pred = self._rpn_head(p)
            all_fpn_scores.append(cur_score)
            all_fpn_bbox_pred.append(cur_bboxes_pred)
        return torch.cat(all_fpn_scores, dim=0), torch.cat(all_fpn_bbox_pred, dim=0)

    def _fg_scores(self, all_fpn_scores):
        """tf.nn.softmax(all_fpn_scores)[:, 1] (:223) -- the HIP kernel, bit-identical to the fused proposal stage"""
        return ops.rpn_fg_softmax(all_fpn_scores, 1, ops.RPN_LAYOUT_FPN)

    # ---- :202-276 ------------------------------------------------------------------------------
    @caller_range_checked
    def forward(self, inputs, training=None, mask=None):
        if training:
            image, gt_bboxes, gt_labels = inputs
        else:
            image = inputs
        image = _image_nhwc(image)
        image_shape = [int(image.shape[1]), int(image.shape[2])]
        with torch.no_grad():
            c_list = self._extractor(image, training=training)
            p_list = self._neck(c_list, training=training)
            all_fpn_scores, all_fpn_bbox_pred = self._get_fpn_head_results(p_list)
            all_anchors = self._get_anchors(image_shape)
            cur_scores = self._fg_scores(all_fpn_scores)
            rois = self._rpn_proposal((all_fpn_bbox_pred, all_anchors, cur_scores, image_shape), training=training)
        if training:
            rpn_labels, rpn_bbox_targets, rpn_in_weights, rpn_out_weights = self._anchor_target(
                (gt_bboxes, image_shape, all_anchors), training)
            rpn_cls_loss, rpn_reg_loss = self._get_rpn_loss(all_fpn_scores, all_fpn_bbox_pred, rpn_labels,
                                                            rpn_bbox_targets, rpn_in_weights, rpn_out_weights)
            final_rois, roi_labels, roi_bbox_target, roi_in_weights, roi_out_weights = self._proposal_target(
                (rois, gt_bboxes, gt_labels), training)
            rois_list, selected_idx = self._assign_levels(final_rois)
            with torch.no_grad():
                roi_features = self._get_roi_features(rois_list, p_list, image_shape)
                roi_score, roi_bboxes_txtytwth = self._roi_head(roi_features, training=training)
            roi_cls_loss, roi_reg_loss = self._get_roi_loss(roi_score, roi_bboxes_txtytwth, roi_labels[selected_idx],
                                                            roi_bbox_target[selected_idx], roi_in_weights[selected_idx],
                                                            roi_out_weights[selected_idx])
            return rpn_cls_loss, rpn_reg_loss, roi_cls_loss, roi_reg_loss
        with torch.no_grad():
            rois_list, _ = self._assign_levels(rois)
            roi_features = self._get_roi_features(rois_list, p_list, image_shape)
            roi_score, roi_bboxes_txtytwth = self._roi_head(roi_features, training=training)
            roi_score_softmax = torch.softmax(roi_score.float(), dim=-1)
            roi_bboxes_txtytwth = roi_bboxes_txtytwth.float().reshape(-1, self.num_classes, 4)
            final_rois = torch.cat(rois_list, dim=0)
            return post_ops_prediction(roi_score_softmax, roi_bboxes_txtytwth, final_rois, image_shape,
                                       self._roi_proposal_means, self._roi_proposal_stds,
                                       max_num_per_class=self._prediction_max_objects_per_class,
                                       max_num_per_image=self._prediction_max_objects_per_image,
                                       nms_iou_threshold=self._prediction_nms_iou_threshold,
                                       score_threshold=self._prediction_score_threshold, extractor_stride=16,
                                       num_classes=self.num_classes)

    call = forward

    def _get_rpn_loss(self, rpn_score, rpn_bbox_txtytwth, anchor_target_labels, anchor_target_bboxes_txtytwth,
                      anchor_target_in_weights, anchor_target_out_weights):
        rpn_selected = torch.nonzero(anchor_target_labels >= 0)[:, 0]                               # :282
        rpn_cls_loss = cls_loss(logits=rpn_score[rpn_selected], labels=anchor_target_labels[rpn_selected])
        rpn_reg_loss = smooth_l1_loss(rpn_bbox_txtytwth, anchor_target_bboxes_txtytwth, anchor_target_in_weights,
                                      anchor_target_out_weights, self._rpn_sigma, dim=[0, 1])
        return rpn_cls_loss, rpn_reg_loss

    def _get_roi_loss(self, roi_score, roi_bbox_txtytwth, proposal_target_labels, proposal_target_bboxes_txtytwth,
                      proposal_target_in_weights, proposal_target_out_weights):
        roi_cls_loss = cls_loss(logits=roi_score, labels=proposal_target_labels)
        roi_reg_loss = smooth_l1_loss(roi_bbox_txtytwth, proposal_target_bboxes_txtytwth, proposal_target_in_weights,
                                      proposal_target_out_weights, sigma=self._roi_sigma)
        return roi_cls_loss, roi_reg_loss

    def _assign_levels(self, all_rois):
        """:303-324 -> (rois per level min_level..max_level, the concatenated row indices): odet_assign_levels (levels,
        stable partition) + one host read of the level counts to cut the lists"""
        k = int(all_rois.shape[0])
        srois, level, perm, counts = ops.assign_levels(all_rois, self._min_level, self._max_level)
        cnt = counts.tolist()
        rois_list, off = [], 0
        for c in cnt:
            rois_list.append(srois[off:off + c])
            off += c
        return rois_list, perm[:k]

    # ---- :326-390 ------------------------------------------------------------------------------
    def predict_rpns(self, image_shape, gt_bboxes):
        all_anchors = self._get_anchors(image_shape)
        rpn_labels, _, _, _ = self._anchor_target((gt_bboxes, image_shape, all_anchors), True)
        return all_anchors[torch.nonzero(rpn_labels > 0)[:, 0]]

    @caller_range_checked
    @torch.no_grad()
    def predict_rois(self, preprocessed_img, gt_bboxes, gt_labels, training=True):
        image = _image_nhwc(preprocessed_img)
        image_shape = [int(image.shape[1]), int(image.shape[2])]
        p_list = self._neck(self._extractor(image, training=training), training=training)
        all_fpn_scores, all_fpn_bbox_pred = self._get_fpn_head_results(p_list)
        all_anchors = self._get_anchors(image_shape)
        rois = self._rpn_proposal((all_fpn_bbox_pred, all_anchors, self._fg_scores(all_fpn_scores), image_shape),
                                  training=training)
        return self._proposal_target((rois, gt_bboxes, gt_labels), True)[0]

    @caller_range_checked
    @torch.no_grad()
    def im_detect(self, preprocessed_img, img_scale):
        image = _image_nhwc(preprocessed_img)
        image_shape = [int(image.shape[1]), int(image.shape[2])]
        p_list = self._neck(self._extractor(image, training=False), training=False)
        all_fpn_scores, all_fpn_bbox_pred = self._get_fpn_head_results(p_list)
        all_anchors = self._get_anchors(image_shape)
        rois = self._rpn_proposal((all_fpn_bbox_pred, all_anchors, self._fg_scores(all_fpn_scores), image_shape),
                                  training=False)
        rois_list, _ = self._assign_levels(rois)
        roi_features = self._get_roi_features(rois_list, p_list, image_shape)
        roi_score, roi_bboxes_txtytwth = self._roi_head(roi_features, training=False)
        new_rois = torch.cat([r for r in rois_list if r.shape[0] != 0], dim=0)
        div = torch.full((1,), float(img_scale), dtype=torch.float32, device=new_rois.device)     # (a true float32 division)
        return torch.softmax(roi_score.float(), dim=-1), roi_bboxes_txtytwth.float(), new_rois / div


class _Part(torch.nn.Module):
    """a dense part of ResnetV1Fpn as a callable layer: `fn(x)` with keras' call signature"""

    def __init__(self, fn):
        super().__init__()
        self._fn = [fn]

    @torch.no_grad()
    def forward(self, inputs, training=None, mask=None):
        return self._fn[0](inputs)

    call = forward


class ResnetV1Fpn(BaseFPN):
    """reference resnet_fpn.py:410-543 (same constructor arguments and defaults).  The dense parts -- extractor
    (get_resnet_v1_extractor :262-289), ResnetFpnNeck (:339-407), the RpnHead's convolutions and ResnetRoiHead (:292-336;
    dropout is inference-only identity) -- are the hand-written kernels of model/fpn_detector.ResNetFpnDetector, which this
    class owns as `dense`; weights are randomly initialised with the reference's initialisers (no checkpoints offline).
    `dtype`: torch.float32 (the reference's precision) or torch.float16 (throughput mode); `f32_form`: the float32 layers'
    arithmetic, 'exact' (float32 matrix instructions), 'x3' (split precision, three bfloat16 limbs) or 'x2' (two float16 limbs; a
    pass that leaves float16's range is repeated on three limbs: fpn_detector.caller_range_checked) -- ops.f32_form."""

    def __init__(self, depth=50, roi_head_keep_dropout_rate=0.5, roi_feature_size=(7, 7, 256), num_classes=21,
                 weight_decay=0.0001, level_name_list=('p2', 'p3', 'p4', 'p5', 'p6'), min_level=2, max_level=5,
                 top_down_dims=256, anchor_stride_list=(4, 8, 16, 32, 64), base_anchor_size_list=(32, 64, 128, 256, 512),
                 ratios=(0.5, 1.0, 2.0), scales=(1.,), rpn_proposal_means=(0, 0, 0, 0),
                 rpn_proposal_stds=(1.0, 1.0, 1.0, 1.0), rpn_proposal_num_pre_nms_train=12000,
                 rpn_proposal_num_post_nms_train=2000, rpn_proposal_num_pre_nms_test=6000,
                 rpn_proposal_num_post_nms_test=1000, rpn_proposal_nms_iou_threshold=0.7, rpn_sigma=3.0,
                 rpn_training_pos_iou_threshold=0.7, rpn_training_neg_iou_threshold=0.3,
                 rpn_training_total_num_samples=256, rpn_training_max_pos_samples=128, roi_proposal_means=(0, 0, 0, 0),
                 roi_proposal_stds=(0.1, 0.1, 0.2, 0.2), roi_pool_size=7, roi_pooling_max_pooling_flag=True, roi_sigma=1,
                 roi_training_pos_iou_threshold=0.5, roi_training_neg_iou_threshold=0.1,
                 roi_training_total_num_samples=256, roi_training_max_pos_samples=64,
                 prediction_max_objects_per_image=50, prediction_max_objects_per_class=50,
                 prediction_nms_iou_threshold=0.3, prediction_score_threshold=0.3, dtype=torch.float32, device='cuda',
                 f32_form='exact'):
        from .fpn_detector import ResNetFpnDetector, check_caller_f32_form
        check_caller_f32_form(f32_form)
        if top_down_dims != 256 or tuple(roi_feature_size) != (roi_pool_size, roi_pool_size, top_down_dims):
            raise ValueError('ResnetV1Fpn: the dense kernels are built for 256 top-down channels and %dx%dx256 RoI features'
                             % (roi_pool_size, roi_pool_size))
        if len(ratios) * len(scales) != 3:
            raise ValueError('ResnetV1Fpn: the RpnHead is built for 3 anchors per cell (ratios x scales)')
        torch.nn.Module.__init__(self)             # (the dense part must exist before BaseFPN.__init__ asks for the layers)
        self._depth = depth
        self._roi_head_keep_dropout_rate = roi_head_keep_dropout_rate
        self._top_down_dims = top_down_dims
        dense = ResNetFpnDetector(depth, num_classes, (64, 64), 1, dtype=dtype, f32_form=f32_form)
        dense.to(device=device, dtype=dtype, memory_format=torch.channels_last).eval()
        self.__dict__['_dense_ref'] = dense
        BaseFPN.__init__(
            self, roi_feature_size=roi_feature_size, num_classes=num_classes, weight_decay=weight_decay,
            level_name_list=level_name_list, min_level=min_level, max_level=max_level, anchor_stride_list=anchor_stride_list,
            base_anchor_size_list=base_anchor_size_list, ratios=ratios, scales=scales, rpn_proposal_means=rpn_proposal_means,
            rpn_proposal_stds=rpn_proposal_stds, rpn_proposal_num_pre_nms_train=rpn_proposal_num_pre_nms_train,
            rpn_proposal_num_post_nms_train=rpn_proposal_num_post_nms_train,
            rpn_proposal_num_pre_nms_test=rpn_proposal_num_pre_nms_test,
            rpn_proposal_num_post_nms_test=rpn_proposal_num_post_nms_test,
            rpn_proposal_nms_iou_threshold=rpn_proposal_nms_iou_threshold, rpn_sigma=rpn_sigma,
            rpn_training_pos_iou_threshold=rpn_training_pos_iou_threshold,
            rpn_training_neg_iou_threshold=rpn_training_neg_iou_threshold,
            rpn_training_total_num_samples=rpn_training_total_num_samples,
            rpn_training_max_pos_samples=rpn_training_max_pos_samples, roi_proposal_means=roi_proposal_means,
            roi_proposal_stds=roi_proposal_stds, roi_pool_size=roi_pool_size,
            roi_pooling_max_pooling_flag=roi_pooling_max_pooling_flag, roi_sigma=roi_sigma,
            roi_training_pos_iou_threshold=roi_training_pos_iou_threshold,
            roi_training_neg_iou_threshold=roi_training_neg_iou_threshold,
            roi_training_total_num_samples=roi_training_total_num_samples,
            roi_training_max_pos_samples=roi_training_max_pos_samples,
            prediction_max_objects_per_image=prediction_max_objects_per_image,
            prediction_max_objects_per_class=prediction_max_objects_per_class,
            prediction_nms_iou_threshold=prediction_nms_iou_threshold, prediction_score_threshold=prediction_score_threshold)
        self.dense = dense

    def _get_roi_head(self):
        return _Part(self._dense_ref.roi_head)

    def _get_extractor(self):
        return _Part(self._dense_ref.extractor)

    def _get_neck(self):
        return _Part(self._dense_ref.neck)

    def _get_rpn_head(self, weight_decay):
        return RpnHead(num_anchors=self._num_anchors, weight_decay=weight_decay, dense=self._dense_ref)

    def _get_roi_features(self, rois_list, p_list, image_shape):
        # the dense kernels keep maps channels_last [B,C,H,W]; the pooling layer takes the NHWC view of the same memory
        nhwc = []
        for p in p_list:
            v = p.permute(0, 2, 3, 1)
            nhwc.append(v if v.is_contiguous() else v.contiguous())
        return super()._get_roi_features(rois_list, nhwc, image_shape)

"""BaseFasterRcnn / RpnHead / ResNetFasterRcnn / Vgg16FasterRcnn -- counterparts of the reference's
model/faster_rcnn/base_faster_rcnn_model.py (BaseFasterRcnn :16-306, RpnHead :309-350), resnet_faster_rcnn.py
(ResNetFasterRcnn :188-293) and vgg16_faster_rcnn.py (Vgg16FasterRcnn :11-115): the CALLER objects of the single-level hot
path, with the reference's constructor arguments, `call(inputs, training=None, mask=None)` and `im_detect`.

`call` composes the reference-named layers in the reference's order (:126-198): `_extractor` -> `_anchor_generator`
(generate_by_anchor_base_tf) -> `_rpn_head` -> the [A bg | A fg] score re-layout + softmax -> `_rpn_proposal` ->
`_roi_pooling` (RoiPoolingCropAndResize) -> `_roi_head` -> softmax -> post_ops_prediction, with dynamic shapes and one image
per call as in the reference.  The static-shape, sync-free arrangement for throughput is model/frcnn_detector.py."""
import math

import torch

from .. import ops
from ..utils.anchor_generator import generate_anchor_base, generate_by_anchor_base_tf
from .anchor_target import AnchorTarget
from .fpn_detector import caller_range_checked
from .base_fpn_model import _Part, _image_nhwc
from .losses import cls_loss, smooth_l1_loss
from .prediction import post_ops_prediction
from .proposal_target import ProposalTarget
from .region_proposal import RegionProposal
from .roi_pooling import RoiPoolingCropAndResize

__all__ = ['BaseFasterRcnn', 'RpnHead', 'ResNetFasterRcnn', 'Vgg16FasterRcnn']


class RpnHead(torch.nn.Module):
    """reference base_faster_rcnn_model.py:309-350: scores reshaped to [-1, 2A] ([A bg | A fg] per cell), boxes to [-1, 4];
    the convolutions live in `dense` (a detector of model/frcnn_detector.py)."""

    def __init__(self, num_anchors, weight_decay=0.0001, dense=None):
        super().__init__()
        self._num_anchors = num_anchors
        self._dense = [dense]
        if dense is not None and dense.A != num_anchors:
            raise ValueError('RpnHead: %d anchors per cell but the dense part was built for %d' % (num_anchors, dense.A))

    @torch.no_grad()
    def forward(self, inputs, training=None, mask=None):
        scores, deltas = self._dense[0].rpn(inputs)
        return scores.reshape(-1, 2 * self._num_anchors), deltas.reshape(-1, 4)

    call = forward


class BaseFasterRcnn(torch.nn.Module):
    """reference base_faster_rcnn_model.py:16-118 (same constructor arguments, same private attribute names)."""

    def __init__(self, num_classes, weight_decay, ratios, scales, extractor_stride, rpn_proposal_means, rpn_proposal_stds,
                 rpn_proposal_num_pre_nms_train, rpn_proposal_num_post_nms_train, rpn_proposal_num_pre_nms_test,
                 rpn_proposal_num_post_nms_test, rpn_proposal_nms_iou_threshold, rpn_sigma,
                 rpn_training_pos_iou_threshold, rpn_training_neg_iou_threshold, rpn_training_total_num_samples,
                 rpn_training_max_pos_samples, roi_proposal_means, roi_proposal_stds, roi_pool_size,
                 roi_pooling_max_pooling_flag, roi_sigma, roi_training_pos_iou_threshold, roi_training_neg_iou_threshold,
                 roi_training_total_num_samples, roi_training_max_pos_samples, prediction_max_objects_per_image,
                 prediction_max_objects_per_class, prediction_nms_iou_threshold, prediction_score_threshold):
        super().__init__()
        self.num_classes = num_classes
        self.weight_decay = weight_decay
        self._ratios = ratios
        self._scales = scales
        self._num_anchors = len(ratios) * len(scales)
        self._extractor_stride = extractor_stride
        self._rpn_sigma = rpn_sigma
        self._roi_sigma = roi_sigma
        self._roi_proposal_means = roi_proposal_means
        self._roi_proposal_stds = roi_proposal_stds
        self._prediction_max_objects_per_image = prediction_max_objects_per_image
        self._prediction_max_objects_per_class = prediction_max_objects_per_class
        self._prediction_nms_iou_threshold = prediction_nms_iou_threshold
        self._prediction_score_threshold = prediction_score_threshold
        self._anchor_generator = generate_by_anchor_base_tf
        self._anchor_base = generate_anchor_base(extractor_stride, ratios, scales).astype('float32')      # :84 tf.to_float
        self._rpn_head = self._get_rpn_head(weight_decay)
        self._rpn_proposal = RegionProposal(
            num_anchors=self._num_anchors, num_pre_nms_train=rpn_proposal_num_pre_nms_train,
            num_post_nms_train=rpn_proposal_num_post_nms_train, num_pre_nms_test=rpn_proposal_num_pre_nms_test,
            num_post_nms_test=rpn_proposal_num_post_nms_test, nms_iou_threshold=rpn_proposal_nms_iou_threshold,
            target_means=rpn_proposal_means, target_stds=rpn_proposal_stds)
        self._anchor_target = AnchorTarget(
            pos_iou_threshold=rpn_training_pos_iou_threshold, neg_iou_threshold=rpn_training_neg_iou_threshold,
            total_num_samples=rpn_training_total_num_samples, max_pos_samples=rpn_training_max_pos_samples,
            target_means=rpn_proposal_means, target_stds=rpn_proposal_stds)
        self._roi_pooling = RoiPoolingCropAndResize(pool_size=roi_pool_size, max_pooling_flag=roi_pooling_max_pooling_flag)
        self._proposal_target = ProposalTarget(
            num_classes=num_classes, pos_iou_threshold=roi_training_pos_iou_threshold,
            neg_iou_threshold=roi_training_neg_iou_threshold, total_num_samples=roi_training_total_num_samples,
            max_pos_samples=roi_training_max_pos_samples, target_means=roi_proposal_means,
            target_stds=roi_proposal_stds)
        self._extractor = self._get_extractor()
        self._roi_head = self._get_roi_head()

    def _get_roi_head(self):
        raise NotImplementedError

    def _get_extractor(self):
        raise NotImplementedError

    def _get_rpn_head(self, weight_decay):
        return RpnHead(num_anchors=self._num_anchors, weight_decay=weight_decay)

    def _anchors_and_proposals(self, image, training):
        """:130-153 / :279-300: extractor -> anchors -> RpnHead -> fg scores -> RegionProposal"""
        image_shape = [int(image.shape[1]), int(image.shape[2])]
        shared_features = self._extractor(image, training=training)
        anchors = self._anchor_generator(self._anchor_base, self._extractor_stride,
                                         int(math.ceil(image_shape[0] / self._extractor_stride)),
                                         int(math.ceil(image_shape[1] / self._extractor_stride)))
        rpn_score, rpn_bbox_txtytwth = self._rpn_head(shared_features, training=training)
        # :147-151 (reshape / transpose / softmax / transpose / slice): the fg probabilities of the [A bg | A fg] layout
        scores = ops.rpn_fg_softmax(rpn_score, self._num_anchors, ops.RPN_LAYOUT_FRCNN)
        rois = self._rpn_proposal((rpn_bbox_txtytwth, anchors, scores, image_shape), training=training)
        return image_shape, shared_features, anchors, rpn_score, rpn_bbox_txtytwth, rois

    @caller_range_checked
    def forward(self, inputs, training=None, mask=None):
        if training:
            image, gt_bboxes, gt_labels = inputs
        else:
            image = inputs
        image = _image_nhwc(image)
        with torch.no_grad():
            image_shape, shared_features, anchors, rpn_score, rpn_bbox_txtytwth, rois = \
                self._anchors_and_proposals(image, training)
        if training:
            rpn_labels, rpn_bbox_targets, rpn_in_weights, rpn_out_weights = self._anchor_target(
                (gt_bboxes, image_shape, anchors), training)
            rpn_cls_loss, rpn_reg_loss = self._get_rpn_loss(rpn_score, rpn_bbox_txtytwth, rpn_labels, rpn_bbox_targets,
                                                            rpn_in_weights, rpn_out_weights)
            final_rois, roi_labels, roi_bbox_target, roi_in_weights, roi_out_weights = self._proposal_target(
                (rois, gt_bboxes, gt_labels), training)
            with torch.no_grad():
                roi_features = self._roi_pooling((shared_features, final_rois, self._extractor_stride), training=training)
                roi_score, roi_bboxes_txtytwth = self._roi_head(roi_features, training=training)
            roi_cls_loss, roi_reg_loss = self._get_roi_loss(roi_score, roi_bboxes_txtytwth, roi_labels, roi_bbox_target,
                                                            roi_in_weights, roi_out_weights)
            return rpn_cls_loss, rpn_reg_loss, roi_cls_loss, roi_reg_loss
        with torch.no_grad():
            roi_features = self._roi_pooling((shared_features, rois, self._extractor_stride), training=training)
            roi_score, roi_bboxes_txtytwth = self._roi_head(roi_features, training=training)
            roi_score_softmax = torch.softmax(roi_score.float(), dim=-1)
            roi_bboxes_txtytwth = roi_bboxes_txtytwth.float().reshape(-1, self.num_classes, 4)
            return post_ops_prediction(roi_score_softmax, roi_bboxes_txtytwth, rois, image_shape,
                                       self._roi_proposal_means, self._roi_proposal_stds,
                                       max_num_per_class=self._prediction_max_objects_per_class,
                                       max_num_per_image=self._prediction_max_objects_per_image,
                                       nms_iou_threshold=self._prediction_nms_iou_threshold,
                                       score_threshold=self._prediction_score_threshold,
                                       extractor_stride=self._extractor_stride, num_classes=self.num_classes)

    call = forward

    def _get_rpn_loss(self, rpn_score, rpn_bbox_txtytwth, anchor_target_labels, anchor_target_bboxes_txtytwth,
                      anchor_target_in_weights, anchor_target_out_weights):
        """:200-215: the [A bg | A fg] scores re-laid as [N, 2] rows before the selection"""
        rpn_score = rpn_score.reshape(-1, 2, self._num_anchors).permute(0, 2, 1).reshape(-1, 2)
        rpn_selected = torch.nonzero(anchor_target_labels >= 0)[:, 0]
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

    def predict_rpn(self, image, gt_bboxes):
        """:226-241 (a debugging helper): the anchors AnchorTarget labels positive for `gt_bboxes`.  The reference's body is
        stale against its own AnchorTarget (it passes (anchors, gt_bboxes, image_shape) and unpacks an index list and a count,
        model/anchor_target.py:37-107 takes (gt_bboxes, image_shape, anchors) and returns labels); this is its intent, written
        like BaseFPN.predict_rpns (model/fpn/base_fpn_model.py:326-340)."""
        image = _image_nhwc(image)
        image_shape = [int(image.shape[1]), int(image.shape[2])]
        anchors = self._anchor_generator(self._anchor_base, self._extractor_stride,
                                         int(math.ceil(image_shape[0] / self._extractor_stride)),
                                         int(math.ceil(image_shape[1] / self._extractor_stride)))
        rpn_labels, _, _, _ = self._anchor_target((gt_bboxes, image_shape, anchors), True)
        return anchors[torch.nonzero(rpn_labels > 0)[:, 0]]

    @torch.no_grad()
    def predict_roi(self, image, gt_bboxes, gt_labels):
        """:243-266: (final_rois, final_labels, final_bbox_targets, bbox_inside_weights, bbox_outside_weights) of the
        training-mode proposals"""
        image = _image_nhwc(image)
        rois = self._anchors_and_proposals(image, True)[5]
        return self._proposal_target((rois, gt_bboxes, gt_labels), True)

    @caller_range_checked
    @torch.no_grad()
    def im_detect(self, preprocessed_image, img_scale):
        """:279-306"""
        image = _image_nhwc(preprocessed_image)
        _, shared_features, _, _, _, rois = self._anchors_and_proposals(image, False)
        roi_features = self._roi_pooling((shared_features, rois, self._extractor_stride), training=False)
        roi_score, roi_bboxes_txtytwth = self._roi_head(roi_features, training=False)
        div = torch.full((1,), float(img_scale), dtype=torch.float32, device=rois.device)
        return torch.softmax(roi_score.float(), dim=-1), roi_bboxes_txtytwth.float(), rois / div


def _features_nhwc(dense):
    """the detector's extractor output as the NHWC map RoiPoolingCropAndResize / RpnHead take"""
    def fn(image):
        return dense.features(image)
    return fn


class _FrcnnFromDense(BaseFasterRcnn):
    """shared plumbing of the two concrete models: the dense parts are a detector of model/frcnn_detector.py (`dense`)"""

    def _init_with_dense(self, dense, dtype, device, kw):
        torch.nn.Module.__init__(self)
        dense.to(device=device, dtype=dtype, memory_format=torch.channels_last).eval()
        self.__dict__['_dense_ref'] = dense
        BaseFasterRcnn.__init__(self, **kw)
        self.dense = dense

    def _get_extractor(self):
        return _Part(self._dense_ref.features)

    def _get_roi_head(self):
        return _Part(self._dense_ref.roi_head)

    def _get_rpn_head(self, weight_decay):
        return RpnHead(num_anchors=self._num_anchors, weight_decay=weight_decay, dense=self._dense_ref)

    def _anchors_and_proposals(self, image, training):
        out = list(super()._anchors_and_proposals(image, training))
        # the detectors keep maps channels_last [B,C,H,W]; the pooling layer takes the NHWC view of the same memory
        feat = out[1].permute(0, 2, 3, 1)
        out[1] = feat if feat.is_contiguous() else feat.contiguous()
        return tuple(out)


_COMMON = dict(num_classes=21, weight_decay=0.0001, ratios=(0.5, 1.0, 2.0), scales=(8, 16, 32), extractor_stride=16,
               rpn_proposal_means=(0, 0, 0, 0), rpn_proposal_stds=(1.0, 1.0, 1.0, 1.0),
               rpn_proposal_num_pre_nms_train=12000, rpn_proposal_num_post_nms_train=2000,
               rpn_proposal_num_pre_nms_test=6000, rpn_proposal_num_post_nms_test=300, rpn_proposal_nms_iou_threshold=0.7,
               rpn_sigma=3.0, rpn_training_pos_iou_threshold=0.7, rpn_training_neg_iou_threshold=0.3,
               rpn_training_total_num_samples=256, rpn_training_max_pos_samples=128, roi_proposal_means=(0, 0, 0, 0),
               roi_proposal_stds=(0.1, 0.1, 0.2, 0.2), roi_pool_size=7, roi_pooling_max_pooling_flag=True, roi_sigma=1,
               roi_training_pos_iou_threshold=0.5, roi_training_neg_iou_threshold=0.1, roi_training_total_num_samples=128,
               roi_training_max_pos_samples=32, prediction_max_objects_per_image=50, prediction_max_objects_per_class=50,
               prediction_nms_iou_threshold=0.3, prediction_score_threshold=0.3)


class ResNetFasterRcnn(_FrcnnFromDense):
    """reference resnet_faster_rcnn.py:188-293: ResNet-{50,101,152} C4 extractor, RoI head = conv5 stack + global average
    pooling + score / box layers.  Keyword arguments and defaults as there (`_COMMON`; model_factory.py:117 passes
    roi_pooling_max_pooling_flag=False from the config)."""

    def __init__(self, depth=50, roi_feature_size=(7, 7, 1024), dtype=torch.float32, device='cuda', f32_form='exact', **kwargs):
        from .frcnn_detector import ResNetC4Detector
        from .fpn_detector import check_caller_f32_form
        check_caller_f32_form(f32_form)
        if depth not in [50, 101, 152]:
            raise ValueError('unknown resnet layers number {}'.format(depth))
        kw = dict(_COMMON)
        unknown = set(kwargs) - set(kw)
        if unknown:
            raise TypeError('unexpected keyword argument(s) %s' % sorted(unknown))
        kw.update(kwargs)
        self._depth = depth
        self._roi_feature_size = roi_feature_size
        if len(kw['ratios']) * len(kw['scales']) != 9 or kw['extractor_stride'] != 16:
            raise ValueError('ResNetFasterRcnn: the dense kernels are built for 9 anchors per cell at stride 16')
        self._init_with_dense(ResNetC4Detector(depth, kw['num_classes'], (64, 64), 1, dtype=dtype, f32_form=f32_form), dtype, device, kw)


class Vgg16FasterRcnn(_FrcnnFromDense):
    """reference vgg16_faster_rcnn.py:11-115: Vgg16Extractor (:260-342) + Vgg16RoiHead (:178-257; dropout is identity at
    inference).  `slim_ckpt_file_path` must be None: no checkpoint exists offline (random initialisation)."""

    def __init__(self, slim_ckpt_file_path=None, roi_head_keep_dropout_rate=0.5, roi_feature_size=(7, 7, 512),
                 dtype=torch.float32, device='cuda', f32_form='exact', **kwargs):
        from .frcnn_detector import Vgg16Detector
        from .fpn_detector import check_caller_f32_form
        check_caller_f32_form(f32_form)
        if slim_ckpt_file_path is not None:
            raise ValueError('Vgg16FasterRcnn: checkpoint import is out of scope (SURVEY section 2); weights are random')
        kw = dict(_COMMON)
        unknown = set(kwargs) - set(kw)
        if unknown:
            raise TypeError('unexpected keyword argument(s) %s' % sorted(unknown))
        kw.update(kwargs)
        self._roi_head_keep_dropout_rate = roi_head_keep_dropout_rate
        self._roi_feature_size = roi_feature_size
        if len(kw['ratios']) * len(kw['scales']) != 9 or kw['extractor_stride'] != 16:
            raise ValueError('Vgg16FasterRcnn: the dense kernels are built for 9 anchors per cell at stride 16')
        self._init_with_dense(Vgg16Detector(kw['num_classes'], (64, 64), 1, dtype=dtype, f32_form=f32_form), dtype, device, kw)

"""ResNet-C4 Faster R-CNN detector assembled around the HIP hot path -- counterpart of the reference's
model/faster_rcnn/resnet_faster_rcnn.py (ResNetFasterRcnn: extractor conv1..conv4, RoI head = conv5 stack +
global average pool + two dense layers) + the inference branch of model/faster_rcnn/base_faster_rcnn_model.py
(BaseFasterRcnn.call :126-198, RpnHead :309-350).  BASELINE config "ResNet-50 Faster R-CNN, 1x3x800x1333".

Same arrangement as model/fpn_detector.py: the convolutions are genuine dense contractions and run on the matrix cores
through this repository's kernels (NHWC, float32 = parity mode / float16 = throughput mode; no library convolution or GEMM
route, no CPU path); everything between them is FrcnnHotPath (anchors in registers -> [A bg | A fg] softmax -> decode /
clip -> exact NMS over all anchors -> 7x7 crop (ResNet) / 14x14 crop + 2x2 max (VGG16) on the stride-16 map ->
post_ops_prediction).  Weights are randomly initialised with the reference's initialisers (no checkpoints offline), frozen
batch-norm folded.  The plain-torch formulation of the same networks lives with the tests (tests/torch_reference.py)."""
import torch
import torch.nn as nn

from .. import ops
from ..pipeline import FrcnnHotPath, FrcnnStepBatch
from . import fpn_detector as fpn
from .fpn_detector import _BLOCKS, DEFAULT_BLIND_CHUNKS, ResNetFpnDetector, _NmsCompleteness, _FinalLayer, _conv, _conv_epi, _stem, \
    _conv_relu_pool, _fold_frozen_bn, _in_f32_form, _nhwc, _no_kernel, _stack, rpn_pair_weights

__all__ = ['ResNetC4Detector', 'Vgg16Detector']


class ResNetC4Detector(_NmsCompleteness, _FinalLayer, nn.Module):
    """Inference-only ResNet-{50,101,152} C4 Faster R-CNN.  `forward(images)` takes NHWC float images [B,H,W,3]
    (mean-subtracted) and returns, per image, the padded detections of post_ops_prediction + their count."""

    def __init__(self, depth=50, num_classes=21, image_shape=(800, 1333), num_proposals=300, dtype=torch.float32,
                 max_batch=1, roi_chunk=0, f32_form='exact', **hot_kwargs):
        super().__init__()
        b = _BLOCKS[depth]
        self.dtype = dtype
        self.f32_form = f32_form             # float32 mode: 'exact' | 'x3' | 'x2' (model/fpn_detector.py)
        self.image_shape = (int(image_shape[0]), int(image_shape[1]))
        self.num_classes = num_classes
        # extractor (resnet_faster_rcnn.py:104-153): conv1 .. conv4, stride 16
        self.conv1 = _fold_frozen_bn(_conv(3, 64, 7, 2, 0))
        self.conv2 = _stack(64, 64, b[0], 1)
        self.conv3 = _stack(256, 128, b[1], 2)
        self.conv4 = _stack(512, 256, b[2], 2)
        # RPN head (base_faster_rcnn_model.py:309-350); 9 anchors per cell, scores laid out [A bg | A fg]
        self.A = 9
        self.rpn_conv = _conv(1024, 512, 3, 1, 1, std=0.01)
        self.rpn_score = _conv(512, 2 * self.A, 1, std=0.01)
        self.rpn_bbox = _conv(512, 4 * self.A, 1, std=0.001)
        # RoI head (resnet_faster_rcnn.py:156-183): conv5 stack (stride1 = 1) on the 7x7 crops, GAP, 2 dense
        self.conv5 = _stack(1024, 512, b[3], 1)
        self.score = nn.Linear(2048, num_classes)
        self.bbox = nn.Linear(2048, 4 * num_classes)
        for m, std in ((self.score, 0.01), (self.bbox, 0.001)):
            nn.init.normal_(m.weight, 0.0, std)
            nn.init.zeros_(m.bias)
        self._hot_args = (self.image_shape, num_classes, num_proposals, 1024)
        # config/faster_rcnn_config.py: 'resnet_roi_pooling_max_pooling_flag': False (7x7 crop, no pool) -- what
        # model_factory.py:117 passes, overriding the class default
        self._hot_kwargs = dict(pool_size=7, max_pooling_flag=False, blind_chunks=DEFAULT_BLIND_CHUNKS)
        self._hot_kwargs.update(hot_kwargs)
        self._hot = []
        self._max_batch = max_batch
        self._roi_chunk = int(roi_chunk)

    def prepare(self, device='cuda'):
        """Moves the model to the GPU and allocates the hot path.  Up to 8 images go through the hot path in the SAME
        kernel launches (FrcnnStepBatch: odet_fpn_step_t.single_level) and through the RoI head as one batch;
        `batched=False` in the hot-path keywords selects one FrcnnHotPath per image on a stream of its own."""
        self.to(device=device, dtype=self.dtype, memory_format=torch.channels_last).eval()
        ops.invalidate_planes(self)                   # (cached limb planes of weights that may have been rewritten through .data)
        # float16 maps go straight into the RoI kernel (pooled 14x14 + max and un-pooled 7x7 crop alike)
        feat_dtype = torch.float16 if self.dtype == torch.float16 else torch.float32
        self._feature_dtype = feat_dtype
        kw = dict(self._hot_kwargs)
        self._steps = None
        if self._max_batch <= 64 and kw.pop('batched', True):
            self._steps = FrcnnStepBatch(self._max_batch, *self._hot_args, feature_dtype=feat_dtype, **kw)
            self._hot = self._steps.slots
            self._roi_feat_all = self._steps.roi_features
            K = self._hot_args[2]
            dev = self._hot[0].device
            self._cls = torch.zeros((self._max_batch, K, self.num_classes), dtype=torch.float32, device=dev)
            self._dlt = torch.zeros((self._max_batch, K, 4 * self.num_classes), dtype=torch.float32, device=dev)
            self._bound = False
            return self
        self._hot = [FrcnnHotPath(*self._hot_args, feature_dtype=feat_dtype, **kw) for _ in range(self._max_batch)]
        # the images' RoI features are consecutive blocks of one buffer: the RoI head takes the whole batch at once
        h0 = self._hot[0]
        self._roi_feat_all = torch.zeros((self._max_batch,) + tuple(h0.roi_features.shape), dtype=feat_dtype, device=h0.device)
        for b, h in enumerate(self._hot):
            h.roi_features = self._roi_feat_all[b]
        # one stream per image: the hot path of an image is a chain of small launches (~130 us), the images'
        # chains run beside each other
        self._streams = [torch.cuda.Stream(device=h0.device) for _ in range(self._max_batch)]
        return self

    def _per_image(self, B, fn):
        """fn(b) for every image on its own stream; the current stream forks before and joins after (events, no host
        sync -- capturable into a HIP graph).  The tensors fn touches stay referenced by the caller past the join."""
        if B == 1:
            return [fn(0)]
        cur = torch.cuda.current_stream()
        fork = torch.cuda.Event()
        fork.record(cur)
        out = []
        for b in range(B):
            st = self._streams[b]
            st.wait_event(fork)
            with torch.cuda.stream(st):
                out.append(fn(b))
            done = torch.cuda.Event()
            done.record(st)
            cur.wait_event(done)
        return out

    # ---- dense parts ---------------------------------------------------------------------------
    @_in_f32_form
    def features(self, images_nhwc):
        """[B,H,W,3] -> C4 [B,1024,ceil(H/16),ceil(W/16)] channels_last (= NHWC in memory)."""
        # conv1_pad + valid 7x7/2, bias + ReLU, pool1_pad (zeros) + 3x3/2 (float16: one launch from the image)
        x = _stem(self.conv1, images_nhwc, self.dtype)
        return self.conv4(self.conv3(self.conv2(x)))

    @_in_f32_form
    def rpn(self, c4):
        """RpnHead: scores [B, fh*fw, 2A] ([A bg | A fg] per location), deltas [B, fh*fw*A, 4] (float32)."""
        x = _conv_epi(self.rpn_conv, c4, relu=True)
        B = x.shape[0]
        gran = 64 if x.dtype == torch.float16 else 32
        if x.shape[1] % gran or x.shape[1] < 2 * gran:
            raise _no_kernel('RpnHead 1x1 pair', self.rpn_score, x)
        # the two 1x1 convolutions as ONE contraction on the pointwise GEMM kernel (weight rows zero-padded to 64), then ONE
        # pass: + bias, float32, split (ops.rpn_pack_pair; [fh*fw, 2A] and [fh*fw*A, 2] are the same memory)
        w, b = rpn_pair_weights(self)
        sd = ops.pointwise(_nhwc(x), fpn.rpn_pair_padded(self, w), None)[..., :6 * self.A]
        n = int(sd.shape[1]) * int(sd.shape[2]) * self.A
        scores = torch.empty((B, n, 2), dtype=torch.float32, device=x.device)
        deltas = torch.empty((B, n, 4), dtype=torch.float32, device=x.device)
        ops.rpn_pack_pair(sd if sd.is_contiguous() else sd.contiguous(), b, self.A, scores, deltas, 0)
        return scores.view(B, -1, 2 * self.A), deltas

    def head_activation(self, roi_features):
        """[R,7,7,1024] NHWC -> conv5 -> global average pool [R,2048] (resnet_faster_rcnn.py:156-183): the input of the score /
        bbox layers"""
        x = roi_features.permute(0, 3, 1, 2).to(self.dtype)                      # NCHW view of NHWC memory
        R = x.shape[0]
        step = self._roi_chunk if self._roi_chunk > 0 else R
        outs = [self.conv5(x[i:i + step]).mean(dim=(2, 3)) for i in range(0, R, step)]
        y = outs[0] if len(outs) == 1 else torch.cat(outs, 0)
        return y.contiguous()

    @_in_f32_form
    def roi_head(self, roi_features):
        """-> (score logits [R,C], box deltas [R,4C]), float32"""
        return self._final_outputs(self.head_activation(roi_features))

    capture = ResNetFpnDetector.capture          # whole forward pass as one HIP graph (generic over self.forward)

    # ---- the model ----------------------------------------------------------------------------------
    def _run_to_head(self, images_nhwc):
        """extractor -> RPN head -> proposals -> RoI features -> RoI head (base_faster_rcnn_model.py:132-187 /
        :279-304): per image (class softmax [K,Ccls], raw deltas [K,4*Ccls]); rows >= the proposal count are padding."""
        B = images_nhwc.shape[0]
        if B > len(self._hot):
            raise ValueError('batch %d exceeds max_batch %d' % (B, len(self._hot)))
        c4 = self.features(images_nhwc)
        rpn_scores, rpn_deltas = self.rpn(c4)
        rpn_scores, rpn_deltas = rpn_scores.float().contiguous(), rpn_deltas.float().contiguous()
        maps = c4.permute(0, 2, 3, 1)                                            # NHWC view
        if maps.dtype != self._feature_dtype:
            maps = maps.to(self._feature_dtype)
        maps = maps.contiguous()
        K = self._roi_feat_all.shape[1]
        if self._steps is not None:
            # B images in the same hot-path launches, the RoI head on all B x K crops at once
            sb = self._steps
            bind = sb.rebind if self._bound else sb.bind
            for b in range(B):
                bind(b, rpn_scores[b], rpn_deltas[b], [maps[b:b + 1]], self._cls[b], self._dlt[b])
            if B == self._max_batch:
                self._bound = True
            sb.enqueue(sb.STAGE_PROPOSALS | sb.STAGE_ROI, B)
            feats = sb.roi_features[:B].reshape((B * K,) + tuple(sb.roi_features.shape[2:]))
            logits, bbox = self.roi_head(feats)
            torch.softmax(logits.float(), dim=-1, out=self._cls[:B].view(B * K, -1))
            self._dlt[:B].view(B * K, -1).copy_(bbox)
            heads = [(self._cls[b], self._dlt[b]) for b in range(B)]
            self._last_pass = (rpn_scores, rpn_deltas, maps, heads)
            return heads
        def proposals_and_crops(b):
            hot = self._hot[b]
            hot.stage_proposals(rpn_scores[b], rpn_deltas[b])
            hot.stage_roi(maps[b:b + 1])
        self._per_image(B, proposals_and_crops)
        feats = self._roi_feat_all[:B].reshape((B * K,) + tuple(self._roi_feat_all.shape[2:]))
        logits, bbox = self.roi_head(feats)                                      # one head pass for the whole batch
        cls = torch.softmax(logits.float(), dim=-1).reshape(B, K, -1).contiguous()
        bbox = bbox.float().reshape(B, K, -1).contiguous()
        heads = [(cls[b], bbox[b]) for b in range(B)]
        self._last_pass = (rpn_scores, rpn_deltas, maps, heads)
        return heads

    @staticmethod
    def _maps_of(maps, b):
        return maps[b:b + 1]

    @torch.no_grad()
    def forward(self, images_nhwc, check=None):
        def run(im):
            heads = self._run_to_head(im)
            B = len(heads)
            if self._steps is not None:
                sb = self._steps
                sb.enqueue(sb.STAGE_DETECT, B)
                return [(h.det_boxes, h.det_labels, h.det_scores, h.det_count) for h in sb.slots[:B]]
            return self._per_image(B, lambda b: self._hot[b].stage_detect(heads[b][0], heads[b][1]))
        return self._forward_checked(images_nhwc, check, run)

    @torch.no_grad()
    def im_detect(self, images_nhwc, img_scale):
        """The evaluation entry of the reference models (base_faster_rcnn_model.py:279-306): per image
        (softmax scores [R,Ccls], raw deltas [R,4*Ccls], rois / img_scale [R,4]) for the R proposals the image kept
        (NMS order); consumed by evaluation.pascal_eval.detect_image with img_scale = 1.  Host-syncs once."""
        heads = self.run_range_checked(lambda: self._run_to_head(images_nhwc))     # ('x2': out of range -> again on three limbs)
        B = len(heads)
        self._last_batch = B
        self.recover(B)
        out = []
        for b, (cls, dlt) in enumerate(heads):
            hot = self._hot[b]
            k = int(hot.roi_count.item())
            sc = img_scale[b] if isinstance(img_scale, (list, tuple)) or (hasattr(img_scale, 'ndim') and img_scale.ndim > 0) else img_scale
            div = torch.full((1,), float(sc), dtype=torch.float32, device=hot.rois.device)   # (true division: see the FPN detector)
            out.append((cls[:k].clone(), dlt[:k].clone(), hot.rois[:k] / div))
        return out


class Vgg16Detector(ResNetC4Detector):
    """Inference-only VGG16 Faster R-CNN -- counterpart of model/faster_rcnn/vgg16_faster_rcnn.py
    (Vgg16Extractor :260-342: 13 3x3 'same' convolutions + ReLU, four 2x2/2 'same' max-pools, stride 16;
    Vgg16RoiHead :178-257: flatten(7,7,512) -> fc 4096 -> fc 4096 -> score / boxes) around FrcnnHotPath
    (14x14 crop + 2x2 max on 512 channels).  BASELINE config "VGG16 Faster R-CNN, 1x3x600x800"."""

    _CFG = ((64, 2), (128, 2), (256, 3), (512, 3), (512, 3))

    def __init__(self, num_classes=21, image_shape=(600, 800), num_proposals=300, dtype=torch.float32, max_batch=1,
                 f32_form='exact', **hot_kwargs):
        nn.Module.__init__(self)
        self.dtype = dtype
        self.f32_form = f32_form
        self.image_shape = (int(image_shape[0]), int(image_shape[1]))
        self.num_classes = num_classes
        convs, cin = [], 3
        for cout, n in self._CFG:
            for _ in range(n):
                convs.append(_conv(cin, cout, 3, 1, 1))
                cin = cout
        self.convs = nn.ModuleList(convs)
        self.A = 9
        self.rpn_conv = _conv(512, 512, 3, 1, 1, std=0.01)
        self.rpn_score = _conv(512, 2 * self.A, 1, std=0.01)
        self.rpn_bbox = _conv(512, 4 * self.A, 1, std=0.001)
        self.fc1 = nn.Linear(7 * 7 * 512, 4096)
        self.fc2 = nn.Linear(4096, 4096)
        self.score = nn.Linear(4096, num_classes)
        self.bbox = nn.Linear(4096, 4 * num_classes)
        for m, std in ((self.fc1, 0.01), (self.fc2, 0.01), (self.score, 0.01), (self.bbox, 0.001)):
            nn.init.normal_(m.weight, 0.0, std)
            nn.init.zeros_(m.bias)
        self._hot_args = (self.image_shape, num_classes, num_proposals, 512)
        self._hot_kwargs = dict(pool_size=7, max_pooling_flag=True, blind_chunks=DEFAULT_BLIND_CHUNKS)
        self._hot_kwargs.update(hot_kwargs)
        self._hot = []
        self._max_batch = max_batch
        self._roi_chunk = 0

    @_in_f32_form
    def features(self, images_nhwc):
        """[B,H,W,3] -> conv5_3 [B,512,ceil(H/16),ceil(W/16)] channels_last."""
        first = self.convs[0]
        if not images_nhwc.is_cuda or not images_nhwc.is_contiguous() or first.out_channels != 64:
            raise _no_kernel('first convolution', first, images_nhwc)
        if self.dtype == torch.float16 and images_nhwc.dtype in (torch.float32, torch.float16):
            # conv1_1 straight from the image (ops.conv3x3_rgb: bias + ReLU in the launch, its output written once)
            key = (first.weight.data_ptr(), first.weight._version)
            packed = getattr(first, '_odet_packed', None)
            if packed is None or packed[0] != key:
                packed = (key, ops.conv3x3_rgb_pack_weights(first.weight))
                first._odet_packed = packed
            x = ops.conv3x3_rgb(images_nhwc, packed[1], first.bias, relu=True).permute(0, 3, 1, 2)
        elif self.dtype == torch.float32 and images_nhwc.dtype == torch.float32:
            # float32 (parity mode): conv1_1 as the exact-float32 GEMM on its patch matrix (ops.rgb_patches3x3_f32)
            key = (first.weight.data_ptr(), first.weight._version)
            packed = getattr(first, '_odet_packed32', None)
            if packed is None or packed[0] != key:
                with torch.no_grad():
                    w = torch.zeros((64, 64), dtype=torch.float32, device=first.weight.device)
                    w[:, :27] = first.weight.permute(0, 2, 3, 1).reshape(64, 27)
                packed = (key, w)
                first._odet_packed32 = packed
            # (32-bit byte offsets into the patch matrix: groups of images that keep it below 4 GiB)
            Bn, Hn, Wn = (int(v) for v in images_nhwc.shape[:3])
            step = max(1, min(Bn, fpn._PATCH_BYTES_MAX // (Hn * Wn * 64 * 4)))
            parts = [ops.pointwise(ops.rgb_patches3x3_f32(images_nhwc[i:i + step]), packed[1], first.bias, None, True)
                     for i in range(0, Bn, step)]
            x = (parts[0] if len(parts) == 1 else torch.cat(parts, 0)).permute(0, 3, 1, 2)
        else:
            raise _no_kernel('first convolution', first, images_nhwc)
        i = 0
        for bi, (_, n) in enumerate(self._CFG):
            for k in range(n):
                if i == 0:                                    # (conv1_1: above)
                    i += 1
                    continue
                if k == n - 1 and bi < 4:
                    # the stage's last convolution: bias + ReLU + MaxPooling2D((2,2), 2, padding='same') in one pass
                    x = _conv_relu_pool(self.convs[i], x, 2, 2, ceil_mode=True)
                else:
                    x = _conv_epi(self.convs[i], x, relu=True)
                i += 1
        return x

    def head_activation(self, roi_features):
        x = roi_features.reshape(roi_features.shape[0], -1).to(self.dtype)        # Flatten() of NHWC crops
        # fc6 / fc7 on the pointwise GEMM kernel, bias + ReLU in its epilogue (dropout: inference)
        x = ops.dense(x if x.is_contiguous() else x.contiguous(), self.fc1.weight, self.fc1.bias, relu=True)
        return ops.dense(x, self.fc2.weight, self.fc2.bias, relu=True)

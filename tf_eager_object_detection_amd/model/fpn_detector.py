"""ResNet-FPN detector assembled around the HIP hot path -- counterpart of the reference's
model/fpn/resnet_fpn.py (ResnetV1Fpn: extractor, ResnetFpnNeck, ResnetRoiHead) + the inference
branch of model/fpn/base_fpn_model.py (BaseFPN.call, RpnHead).

SURVEY.md section 8(f) ranks 2-3 ("next"): the dense conv / FC stacks are genuine dense contractions and run on the
matrix cores through THIS REPOSITORY'S kernels -- there is no library convolution or GEMM route in this module, at any
batch size, and no CPU path (a layer no kernel takes raises):

  3x3 convolutions   ops.conv3x3_f16 / conv3x3_f32 (implicit GEMM; the ring form at batch 1-2 on the small maps), with the
                     block's last 1x1 convolution + shortcut + ReLU in the same launch where the map is large enough
                     (ops.conv3x3_conv1x1_f16), with the RpnHead's two 1x1 convolutions in the launch (ops.rpn_head_fused)
  1x1 / dense        ops.pointwise / dense (the same kernel with one tap; strided; two sources along K for a stage's first
                     block; the FPN top-down merge in the lateral's epilogue), ops.conv1x1_f16 (register-resident, short K)
  stem               ops.stem_conv7_pool3 (float16: one launch from the image) / the patch-matrix GEMM (float32)

float32 is the parity mode (exact-float32 matrix instructions), float16 the throughput mode.  Weights are randomly
initialised with the reference's initialisers (no checkpoints exist offline); frozen batch-norm (epsilon 1.001e-5,
inference statistics) is folded into the convolutions.  The plain-torch formulation of the same network (library
convolutions; CPU shape bookkeeping and numerical reference) lives with the tests: tests/torch_reference.py.

Shapes follow the reference exactly so that feature maps and anchor grids agree:
conv1 = pad 3 + 7x7/2 valid, pool1 = pad 1 + 3x3/2 valid, the stride of a stage sits on the first
1x1 convolution of its first block, P6 = P5[::2, ::2], top-down merge = 0.5 * resize_bilinear(P_{k+1})
+ 0.5 * lateral with TF1's legacy resize (src = dst * in/out, no half-pixel offset).
"""
import math

import torch
import torch.nn as nn

from .. import ops
from ..pipeline import FpnHotPath, FpnStepBatch

__all__ = ['ResNetFpnDetector', 'tf_legacy_resize_bilinear']

# sync-free NMS chunks of the assembled detectors unless the caller says otherwise: the first two from the ranked
# selection (shared launches), the third on the full order -- enough for clustered (trained-like) and massively tied
# (random-init float16) score distributions; an image that still does not complete is reported empty and flagged
DEFAULT_BLIND_CHUNKS = 4

_BLOCKS = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 152: (3, 8, 36, 3)}
_BN_EPS = 1.001e-5


def _conv(cin, cout, k, stride=1, padding=0, std=None):
    c = nn.Conv2d(cin, cout, k, stride=stride, padding=padding, bias=True)
    if std is None:
        nn.init.kaiming_normal_(c.weight, mode='fan_in', nonlinearity='relu')     # keras 'he_normal'
    else:
        nn.init.normal_(c.weight, 0.0, std)                                       # tf.random_normal_initializer
    nn.init.zeros_(c.bias)
    return c


def _fold_frozen_bn(conv):
    """keras BatchNormalization(trainable=False)(x, training=False) with fresh statistics
    (gamma 1, beta 0, mean 0, var 1) is a multiplication by 1/sqrt(1 + eps): folded into the conv."""
    s = 1.0 / math.sqrt(1.0 + _BN_EPS)
    with torch.no_grad():
        conv.weight.mul_(s)
        conv.bias.mul_(s)
    return conv


def _no_kernel(what, conv, x):
    return RuntimeError('%s: no kernel of this package takes the layer (kernel %s, stride %s, %d -> %d channels, input %s %s on '
                        '%s); the detectors run float16 / float32 NHWC maps on the GPU and have no library or CPU route'
                        % (what, tuple(conv.kernel_size), tuple(conv.stride), conv.in_channels, conv.out_channels,
                           tuple(x.shape), x.dtype, x.device))


def _nhwc(x):
    """the NHWC memory of a channels_last [B,C,H,W] tensor as a contiguous [B,H,W,C] view"""
    y = x.permute(0, 2, 3, 1)
    return y if y.is_contiguous() else y.contiguous()


def _mfma_ok(conv, x):
    """ops.conv1x1_f16 (register-resident operand, weights staged per 64-channel group) takes this 1x1 stride-1 convolution"""
    return (x.is_cuda and x.dtype == torch.float16 and tuple(conv.kernel_size) == (1, 1) and tuple(conv.stride) == (1, 1)
            and tuple(conv.padding) == (0, 0) and conv.in_channels in (64, 128, 256, 512) and conv.out_channels % 64 == 0)


def _pw_ok(conv, x):
    """the pointwise GEMM kernel takes this 1x1 convolution (stride 1 or 2, no padding)"""
    if not x.is_cuda or x.dtype not in (torch.float16, torch.float32):
        return False
    gran = 64 if x.dtype == torch.float16 else 32          # channels per K-step
    return (tuple(conv.kernel_size) == (1, 1) and tuple(conv.padding) == (0, 0)
            and tuple(conv.stride) in ((1, 1), (2, 2)) and conv.in_channels % gran == 0 and conv.in_channels >= 2 * gran
            and conv.out_channels % 64 == 0)


def _route_1x1(conv, x):
    """'mfma' (ops.conv1x1_f16) or 'pw' (ops.pointwise) for this 1x1 convolution.  A fixed rule (the two kernels differ in
    rounding order, so the route must not depend on timing): the LDS-staged GEMM wherever it applies (K >= 128; every first
    1x1 of a bottleneck, the last one with its shortcut, the neck's P5, strided layers, float32) except for 64-channel
    outputs, which -- like the K = 64 layers the GEMM does not take -- go to the register-resident kernel
    (tools/exp/pointwise_layers.py; round 4, cold L2, inside a HIP graph: conv4's last 1x1 with its shortcut at batch 1 / 4
    12.1 / 20.5 us on the GEMM, 13.3 / 30.9 on the register-resident kernel, tools/r04/small_tiles.py)."""
    mfma, pw = _mfma_ok(conv, x), _pw_ok(conv, x)
    if mfma and (conv.out_channels <= 64 or not pw):
        return 'mfma'
    if pw:
        return 'pw'
    raise _no_kernel('1x1 convolution', conv, x)


def _own_conv3x3(conv, x):
    """the implicit-GEMM kernel (ops.conv3x3_f16 / conv3x3_f32) takes this 3x3 stride-1 'same' convolution: float16 with
    cin % 64 == 0 and cout % 64 == 0, float32 with cin % 32 == 0 and cout % 64 == 0 -- at every map size (the launcher picks
    the workgroup tile: 128 .. 256-pixel slabs, or the 64 x 64 ring form when the map has few pixels)"""
    if not x.is_cuda or x.dtype not in (torch.float16, torch.float32):
        return False
    if tuple(conv.kernel_size) != (3, 3) or tuple(conv.stride) != (1, 1) or tuple(conv.padding) != (1, 1):
        return False
    gran = 64 if x.dtype == torch.float16 else 32
    return conv.in_channels % gran == 0 and conv.out_channels % 64 == 0


# A fused bottleneck tail (3x3 + last 1x1 + shortcut + ReLU in one launch) pays from this many 128-pixel slabs on: its
# workgroup must hold ALL middle channels of its pixels, so a small map makes few workgroups (conv4 at batch 1 / 4: 33 /
# 132 on 256 CUs) and the two launches -- 3x3 with its epilogue (ring tiles at batch 1), then the 1x1 GEMM with the shortcut
# in its epilogue -- win: conv4 30.4 vs 53.0 us at batch 1, 48.5 vs 53.0 at batch 2, 55.5 vs 59.0 at batch 4; at batch 8 (263
# slabs) the fused launch: 84.1 vs 90.7; conv3 (cmid 128) at batch 2 / 263 slabs: 33.9 vs 39.6; conv2 from batch 1 on
# (tools/r04/small_tiles.py --only tail, cold L2, inside a HIP graph; profiles/r04_small_tiles_tail.json)
_FUSED_TAIL_MIN_SLABS = 200

# the float32 mode's patch matrices (stem, VGG16's first convolution) are addressed with 32-bit byte offsets
_PATCH_BYTES_MAX = 0xF0000000


def _stem(conv1, images_nhwc, dtype):
    """conv1_pad + 7x7/2 'valid' + folded BN + ReLU + pool1_pad + 3x3/2 max-pooling (resnet_fpn.py:262-289).  float16: ONE
    launch from the image (ops.stem_conv7_pool3: the 64-channel convolution output, 273 MB at batch 8, never goes to
    memory); float32 (parity mode): the 7x7 / 2 convolution as the exact-float32 GEMM on its patch matrix
    (ops.stem_patches_f32: 160 floats per output pixel), then bias + ReLU + the 3x3 / 2 pooling in one pass."""
    ok = (images_nhwc.is_cuda and images_nhwc.is_contiguous() and conv1.out_channels == 64
          and tuple(conv1.kernel_size) == (7, 7) and tuple(conv1.stride) == (2, 2))
    if ok and dtype == torch.float16 and images_nhwc.dtype in (torch.float32, torch.float16):
        key = (conv1.weight.data_ptr(), conv1.weight._version)
        packed = getattr(conv1, '_odet_packed', None)
        if packed is None or packed[0] != key:
            packed = (key, ops.stem_pack_weights(conv1.weight))
            conv1._odet_packed = packed
        return ops.stem_conv7_pool3(images_nhwc, packed[1], conv1.bias).permute(0, 3, 1, 2)
    if ok and dtype == torch.float32 and images_nhwc.dtype == torch.float32:
        key = (conv1.weight.data_ptr(), conv1.weight._version)
        packed = getattr(conv1, '_odet_packed32', None)
        if packed is None or packed[0] != key:
            with torch.no_grad():
                w = torch.zeros((64, 160), dtype=torch.float32, device=conv1.weight.device)
                w[:, :147] = conv1.weight.permute(0, 2, 3, 1).reshape(64, 147)
            packed = (key, w)
            conv1._odet_packed32 = packed
        # (the patch matrix is addressed with 32-bit byte offsets: images go through in groups that keep it below 4 GiB)
        B, H, W = (int(v) for v in images_nhwc.shape[:3])
        per_image = ((H - 1) // 2 + 1) * ((W - 1) // 2 + 1) * 160 * 4
        step = max(1, min(B, _PATCH_BYTES_MAX // per_image))
        parts = []
        for i in range(0, B, step):
            y = ops.pointwise(ops.stem_patches_f32(images_nhwc[i:i + step]), packed[1], None)
            parts.append(ops.bias_relu_maxpool(y, conv1.bias, 3, 2, 1, False))
        y = parts[0] if len(parts) == 1 else torch.cat(parts, 0)
        return y.permute(0, 3, 1, 2)
    raise _no_kernel('stem', conv1, images_nhwc)


def _conv_relu_pool(conv, x, kernel, stride, pool_pad=0, ceil_mode=False):
    """max_pool(relu(conv3x3(x) + bias)) (vgg16_faster_rcnn.py:260-342).  float16 with the 2x2 / 2 'same' pooling: in the
    convolution's own launch (its pixel order makes a pooling window four neighbouring lanes: the un-pooled map is never
    written); otherwise the convolution without its epilogue, then ONE pass (ops.bias_relu_maxpool) that reads its output
    once and writes the pooled map."""
    if not _own_conv3x3(conv, x) or conv.bias is None:
        raise _no_kernel('3x3 convolution + pooling', conv, x)
    if x.dtype == torch.float16 and kernel == 2 and stride == 2 and pool_pad == 0 and ceil_mode:
        return ops.conv3x3_relu_pool2_f16(_nhwc(x), conv.weight, conv.bias).permute(0, 3, 1, 2)
    yn = (ops.conv3x3_f32 if x.dtype == torch.float32 else ops.conv3x3_f16)(_nhwc(x), conv.weight)
    return ops.bias_relu_maxpool(yn, conv.bias, kernel, stride, pool_pad, ceil_mode).permute(0, 3, 1, 2)


def _conv_epi(conv, x, relu=False, residual=None, extra_bias=None):
    """conv -> + bias (+ extra_bias) (+ residual) -> ReLU in ONE launch of this package's kernels; x / residual / result are
    channels_last [B,C,H,W] tensors (= NHWC in memory)."""
    bias = conv.bias if extra_bias is None else conv.bias + extra_bias
    if tuple(conv.kernel_size) == (1, 1):
        res = None if residual is None else _nhwc(residual)
        xn = _nhwc(x)
        if _route_1x1(conv, x) == 'mfma':
            return ops.conv1x1_f16(xn, conv.weight, bias, res, relu).permute(0, 3, 1, 2)
        return ops.pointwise(xn, conv.weight, bias, res, relu, conv.stride[0]).permute(0, 3, 1, 2)
    if residual is None and _own_conv3x3(conv, x):
        # the implicit GEMM with bias (+ ReLU) in its epilogue
        fn = ops.conv3x3_f32 if x.dtype == torch.float32 else ops.conv3x3_f16
        return fn(_nhwc(x), conv.weight, bias, relu=relu).permute(0, 3, 1, 2)
    raise _no_kernel('convolution', conv, x)


class _Block(nn.Module):
    """reference resnet_fpn.py:154-205 block1 (bottleneck, stride on the first 1x1)."""

    def __init__(self, cin, filters, stride, conv_shortcut):
        super().__init__()
        self.short = _fold_frozen_bn(_conv(cin, 4 * filters, 1, stride)) if conv_shortcut else None
        self.c1 = _fold_frozen_bn(_conv(cin, filters, 1, stride))
        self.c2 = _fold_frozen_bn(_conv(filters, filters, 3, 1, 1))
        self.c3 = _fold_frozen_bn(_conv(filters, 4 * filters, 1))
        # Random initialisation only (there are no checkpoints offline): with fresh batch-norm statistics
        # a he_normal residual branch doubles the activation variance per block and 33 blocks overflow
        # fp16; damping the last conv of the branch keeps the random network's activations O(1).
        with torch.no_grad():
            self.c3.weight.mul_(0.2)

    def _dual_weights(self):
        """[w3 | w_shortcut] along K and b3 + b_shortcut (ops.pointwise_dual), cached until a parameter changes"""
        ps = (self.c3.weight, self.c3.bias, self.short.weight, self.short.bias)
        key = tuple((t._version, t.data_ptr(), t.dtype) for t in ps)
        c = getattr(self, '_dual_cache', None)
        if c is None or c[0] != key:
            with torch.no_grad():
                n = self.c3.out_channels
                w = torch.cat([ps[0].reshape(n, -1), ps[2].reshape(n, -1)], 1).contiguous()
                b = (ps[1].float() + ps[3].float()).to(ps[1].dtype).contiguous()
            c = (key, w, b)
            self._dual_cache = c
        return c[1], c[2]

    def forward(self, x):
        y = _conv_epi(self.c1, x, relu=True)
        if self.short is not None:
            # a stage's first block: the last 1x1 convolution AND the convolutional shortcut + Add + ReLU as ONE contraction
            # over [c2's output | the block's (strided) input] with the weights concatenated along K (ops.pointwise_dual) --
            # the shortcut map is never written or re-read, one launch instead of two
            gran = 64 if x.dtype == torch.float16 else 32
            if (self.c3.in_channels % gran or self.short.in_channels % gran or self.c3.out_channels % 64
                    or tuple(self.short.stride) not in ((1, 1), (2, 2))):
                raise _no_kernel('bottleneck with a convolutional shortcut', self.short, x)
            y = _conv_epi(self.c2, y, relu=True)
            w, b = self._dual_weights()
            return ops.pointwise_dual(_nhwc(y), _nhwc(x), w, b, self.short.stride[0], relu=True).permute(0, 3, 1, 2)
        # Add([shortcut, x]) + ReLU ride on c3's epilogue
        if y.dtype == torch.float16 and _own_conv3x3(self.c2, y) and self.c2.out_channels in (64, 128, 256):
            m = int(y.shape[0]) * int(y.shape[2]) * int(y.shape[3])
            if (m + 127) // 128 >= _FUSED_TAIL_MIN_SLABS:
                # the 3x3 convolution AND the block's last 1x1 convolution + bias + shortcut + ReLU in one launch
                # (ops.conv3x3_conv1x1_f16: the 64 / 128 / 256-channel activation between them stays in LDS): conv4 83 vs
                # 108 us at batch 8 (253 vs 320 at 30), conv3 122 vs 134 (400 vs 477), conv2 198 vs 226 (640-700 vs 787)
                out = ops.conv3x3_conv1x1_f16(_nhwc(y), self.c2.weight, self.c2.bias, self.c3.weight, self.c3.bias,
                                              residual=_nhwc(x), relu=True)
                return out.permute(0, 3, 1, 2)
        y = _conv_epi(self.c2, y, relu=True)
        return _conv_epi(self.c3, y, relu=True, residual=x)


def _stack(cin, filters, blocks, stride1):
    layers = [_Block(cin, filters, stride1, True)]
    for _ in range(blocks - 1):
        layers.append(_Block(4 * filters, filters, 1, False))
    return nn.Sequential(*layers)


def tf_legacy_resize_bilinear(x, out_hw):
    """tf.image.resize_bilinear(x, size) of TF 1.x with align_corners=False (resnet_fpn.py:385-398):
    source coordinate = destination index * (in / out), top/left = floor, bottom/right = min(+1, in-1).
    x: [B,C,H,W] (any memory format)."""
    B, C, H, W = x.shape
    oh, ow = int(out_hw[0]), int(out_hw[1])
    dev = x.device
    ys = torch.arange(oh, device=dev, dtype=torch.float32) * (float(H) / float(oh))
    xs = torch.arange(ow, device=dev, dtype=torch.float32) * (float(W) / float(ow))
    y0 = ys.floor().long().clamp_(max=H - 1)
    x0 = xs.floor().long().clamp_(max=W - 1)
    y1 = (y0 + 1).clamp_(max=H - 1)
    x1 = (x0 + 1).clamp_(max=W - 1)
    wy = (ys - y0.float()).to(x.dtype).view(1, 1, oh, 1)
    wx = (xs - x0.float()).to(x.dtype).view(1, 1, 1, ow)
    top = x[:, :, y0, :]
    bot = x[:, :, y1, :]
    tl, tr = top[:, :, :, x0], top[:, :, :, x1]
    bl, br = bot[:, :, :, x0], bot[:, :, :, x1]
    t = tl + (tr - tl) * wx
    b = bl + (br - bl) * wx
    return t + (b - t) * wy


def rpn_pair_weights(m):
    """[6A, 512, 1, 1] weight and [6A] bias of m.rpn_score and m.rpn_bbox concatenated along the output channel
    (the RpnHead's two 1x1 convolutions as one contraction); cached on the module, rebuilt when either parameter
    was modified (weight loading, an optimiser step) or moved."""
    ps = (m.rpn_score.weight, m.rpn_score.bias, m.rpn_bbox.weight, m.rpn_bbox.bias)
    key = tuple((t._version, t.data_ptr(), t.dtype) for t in ps)
    cached = getattr(m, '_rpn_pair', None)
    if cached is None or cached[0] != key:
        with torch.no_grad():
            w = torch.cat([ps[0], ps[2]], 0).contiguous(memory_format=torch.channels_last)
            b = torch.cat([ps[1], ps[3]], 0).contiguous()
        cached = (key, w, b)
        m._rpn_pair = cached
    return cached[1], cached[2]


def rpn_pair_padded(m, w):
    """the concatenated [6A, cin, 1, 1] RpnHead weight as [64 k, cin] with zero rows (the GEMM kernel's channel granule)"""
    c = getattr(m, '_rpn_pad', None)
    if c is None or c[0] is not w:
        with torch.no_grad():
            rows = (int(w.shape[0]) + 63) // 64 * 64
            wp = torch.zeros((rows, w.shape[1]), dtype=w.dtype, device=w.device)
            wp[:w.shape[0]] = w.reshape(w.shape[0], -1)
        c = (w, wp)
        m._rpn_pad = c
    return c[1]


def check_caller_f32_form(form):
    """the reference-surface caller objects (base_fpn_model.py, base_faster_rcnn_model.py) take every float32 form; with 'x2'
    their call / im_detect / predict_rois read the dense part's range status word after the pass and repeat an out-of-range
    pass on three limbs (run_range_checked below), as the detectors do"""
    if form not in ('exact', 'x3', 'x2'):
        raise ValueError("f32_form must be 'exact', 'x3' or 'x2'")


def caller_range_checked(method):
    """decorator for the caller objects' composed passes (call / im_detect / predict_rois): with the dense part on the two-limb
    form, a pass that reported an out-of-range activation is repeated on three limbs (run_range_checked)"""
    import functools

    @functools.wraps(method)
    def run(self, *args, **kw):
        dense = self.__dict__.get('_dense_ref')
        if dense is None or getattr(dense, 'f32_form', 'exact') != 'x2':
            return method(self, *args, **kw)
        return dense.run_range_checked(lambda: method(self, *args, **kw))
    return run


def _in_f32_form(method):
    """runs a detector's dense method with the float32 layers in the detector's `f32_form` ('exact' | 'x3' | 'x2': ops.f32_form)"""
    import functools

    @functools.wraps(method)
    def run(self, *args, **kw):
        form = getattr(self, 'f32_form', 'exact')
        with ops.f32_form(form, workspace=_x3_workspace_of(self) if form != 'exact' else None):
            return method(self, *args, **kw)
    return run


def _x3_workspace_of(model):
    """the split-precision workspace (split-K tickets / parts, the two-limb form's range status word) a detector instance OWNS:
    every launch of the instance -- eager on any stream, or replayed from a graph captured on a side stream -- uses this one, so
    a graph never shares tickets with another instance's graph and range_ok() always reads the word its own launches set.
    One instance's passes must not run concurrently with each other (they share every activation buffer anyway)."""
    ws = model.__dict__.get('_x3_ws')
    if ws is None:
        dev = next(model.parameters()).device
        if dev.type != 'cuda':
            return None
        ws = ops.X3Workspace(dev)
        model.__dict__['_x3_ws'] = ws
    return ws


class _FinalLayer:
    """The RoI heads' last layer of the three detectors: class logits and box regressions as ONE contraction with the
    concatenated [Ccls + 4 Ccls, K] weights (rows zero-padded to a multiple of 64) on the pointwise GEMM kernel, float32
    results in both modes (float32 accumulation AND no rounding of the result: a float16 logit near 10 is 0.008 coarse,
    1 % of a softmax score)."""

    def _final_layer(self):
        ps = (self.score.weight, self.score.bias, self.bbox.weight, self.bbox.bias)
        key = tuple((t._version, t.data_ptr(), t.dtype) for t in ps)
        c = getattr(self, '_final_cache', None)
        if c is None or c[0] != key:
            with torch.no_grad():
                wc = torch.cat([ps[0], ps[2]], 0)
                bc = torch.cat([ps[1], ps[3]], 0).float()
                rows = (wc.shape[0] + 63) // 64 * 64
                wpad = torch.zeros((rows, wc.shape[1]), dtype=wc.dtype, device=wc.device)
                wpad[:wc.shape[0]] = wc
                b = torch.zeros(rows, dtype=torch.float32, device=wc.device)
                b[:bc.shape[0]] = bc
            gran = 64 if wc.dtype == torch.float16 else 32
            c = (key, wpad if wc.shape[1] % gran == 0 and wc.shape[1] >= 2 * gran else None, b.contiguous())
            self._final_cache = c
        return c[1], c[2]

    def _final_outputs(self, x):
        """x [rows, K] (the head's last activation) -> (class logits [rows, Ccls], box regressions [rows, 4 Ccls])"""
        n1 = self.score.out_features
        n5 = n1 + self.bbox.out_features
        wpad, b32 = self._final_layer()
        if not x.is_cuda or x.dtype not in (torch.float16, torch.float32) or wpad is None:
            raise RuntimeError('RoI head: the last layer needs a float16 / float32 GPU activation with a multiple of %d >= %d '
                               'channels (got %s %s on %s)' % (64 if x.dtype == torch.float16 else 32,
                                                                128 if x.dtype == torch.float16 else 64, tuple(x.shape), x.dtype, x.device))
        x = x if x.is_contiguous() else x.contiguous()
        y = ops.dense(x, wpad, b32) if x.dtype == torch.float32 else ops.dense_f16_out_f32(x, wpad, b32)
        return y[:, :n1], y[:, n1:n5]


class _NmsCompleteness:
    """The detectors run the proposal stage sync-free (no host check between kernels; graph-capturable) with a fixed
    number of NMS chunks.  If an image needs more than those, the hot path reports it EMPTY and flags it
    (nms_done = 0, include/odet.h): `forward()` reads the flags after the last launch of the pass whenever it is not
    being captured into a HIP graph and `check_nms` is on (default), and sends ONLY the flagged images through the pass
    again from the RPN head's outputs with the exact proposal stage (host-checked chunks, as many as the image needs:
    the reference's NMS is always exact, model/region_proposal.py:73-81) -- their outputs are overwritten in place, the
    other images are not touched; `nms_reruns` counts them.  Throughput loops that do not want a host sync per pass set
    `model.check_nms = False` and call `recover()` themselves (after a graph replay too); `check_complete()` raises instead."""

    check_nms = True
    nms_reruns = 0
    _last_batch = 0
    _last_pass = None            # (rpn scores, rpn deltas, maps, heads) of the last pass: what a re-run starts from

    def nms_done(self, batch=None):
        """device int32 flags (1 = complete) of the images of the last pass"""
        n = self._last_batch if batch is None else batch
        return [h.nms_done for h in self._hot[:n]]

    def incomplete(self, batch=None):
        """indices of the images of the last pass whose sync-free NMS did not complete (one device -> host copy)"""
        steps = getattr(self, '_steps', None)
        if steps is not None and hasattr(steps, 'nms_done_all'):
            n = self._last_batch if batch is None else batch
            flags = steps.nms_done_all[:n].tolist()
        else:
            flags = [int(t.item()) for t in self.nms_done(batch)]
        return [b for b, f in enumerate(flags) if f != 1]

    def check_complete(self, batch=None):
        bad = self.incomplete(batch)
        if bad:
            raise RuntimeError('the RPN NMS of image(s) %s did not complete inside blind_chunks = %d sync-free chunks: their '
                               'results are reported EMPTY (recover() re-runs them in the exact mode)'
                               % (bad, self._hot[0].blind_chunks))

    def recover(self, batch=None):
        """re-runs the flagged images of the last pass in the exact mode; -> their indices"""
        bad = self.incomplete(batch)
        if bad and self._last_pass is None:
            raise RuntimeError('image(s) %s incomplete and no pass to re-run them from' % bad)
        for b in bad:
            self._rerun_exact(b)
        self.nms_reruns += len(bad)
        return bad

    def _rerun_exact(self, b):
        """image b of the last pass again from the RPN head's outputs: exact proposals -> RoI features -> RoI head ->
        post-ops, into the buffers the pass handed out"""
        rpn_scores, rpn_deltas, maps, heads = self._last_pass
        if maps is None:
            raise RuntimeError('image %d flagged incomplete after its pass was checked and its maps released' % b)
        hot = self._hot[b]
        hot.stage_proposals(rpn_scores[b], rpn_deltas[b], exact=True)
        feats = hot.stage_roi(self._maps_of(maps, b))
        logits, bbox = self.roi_head(feats)
        cls, dlt = heads[b]
        if cls.shape[0] == logits.shape[0] and cls.is_contiguous() and dlt.is_contiguous():
            torch.softmax(logits.float(), dim=-1, out=cls)
            dlt.view(dlt.shape[0], -1).copy_(bbox)
        else:
            cls, dlt = torch.softmax(logits.float(), dim=-1).contiguous(), bbox.float().contiguous()
            heads[b] = (cls, dlt)
        hot.stage_detect(cls, dlt)

    def _after_pass(self, batch, check):
        """-> False if the pass has to be repeated on the three-limb form (range_ok), else True"""
        self._last_batch = batch
        if check is None:
            check = self.check_nms and not torch.cuda.is_current_stream_capturing()
        if check:
            if getattr(self, 'f32_form', 'exact') == 'x2' and not self.range_ok(batch):
                return False
            self.recover(batch)
            # every image of the pass is complete: nothing is left to re-run, so the pass's pyramid (P2..P5: ~90 MB per image)
            # is released here instead of staying pinned until the next pass ends; the RPN outputs and the heads' outputs stay
            lp = self._last_pass
            if lp is not None:
                self._last_pass = (lp[0], lp[1], None, lp[3])
        return True

    # ---- the two-limb float32 form's RANGE (f32_form = 'x2': float16 limbs).  An activation beyond float16's range becomes an
    # infinite limb and every sum it enters is non-finite before bias / shortcut / ReLU, whatever the weights' signs: the
    # launch's epilogue ORs 1 into the RANGE STATUS word of the instance's workspace (include/odet.h; csrc/conv_f32_common.h) --
    # a flag, not a propagated value: a -inf that a ReLU maps to 0 is still reported.  The word is read where the NMS flags
    # are read; a pass that set it is run again on the three-limb form (bfloat16 limbs: float32's range), counted in
    # `range_reruns`.
    range_reruns = 0

    def range_ok(self, batch=None):
        """True iff no two-limb launch of this instance since the last call met an activation outside float16's range (reads
        and clears the status word: one host sync).  After replays of a capture()d graph the caller calls this itself."""
        ws = self.__dict__.get('_x3_ws')
        return True if ws is None else ws.range_ok()

    def run_range_checked(self, fn):
        """fn() -> result on this instance's float32 form; a two-limb pass that reported an out-of-range activation is repeated
        on three limbs (`range_reruns`).  For composed passes (im_detect, the caller objects' call).  A composed pass may also
        FAIL on the non-finite maps of such a pass (no proposal survives, the reference's own torch.cat / tf.concat of an empty
        list raises): the error is the range's if the status word is set -- then the pass is repeated, else it is the caller's."""
        if getattr(self, 'f32_form', 'exact') != 'x2':
            return fn()
        self.range_ok()                                    # (a word left set by an earlier, unchecked pass is not this pass's)
        try:
            out = fn()
            bad = not self.range_ok()
        except Exception:
            if self.range_ok():
                raise
            bad = True
        if bad:
            self.range_reruns += 1
            self.f32_form = 'x3'
            try:
                out = fn()
            finally:
                self.f32_form = 'x2'
        return out

    def _forward_checked(self, images_nhwc, check, run):
        """run(images) -> outputs, then the after-pass checks; a two-limb pass out of range is repeated on three limbs"""
        outs = run(images_nhwc)
        if not self._after_pass(len(outs), check):
            self.range_reruns += 1
            self.f32_form = 'x3'
            try:
                outs = run(images_nhwc)
                self._after_pass(len(outs), check)
            finally:
                self.f32_form = 'x2'
        return outs


class ResNetFpnDetector(_NmsCompleteness, _FinalLayer, nn.Module):
    """Inference-only ResNet-{50,101,152}-FPN detector.  `forward(images)` takes NHWC float images
    [B,H,W,3] (already mean-subtracted, as the reference's input pipeline delivers them) and returns,
    per image, the padded detections of post_ops_prediction plus their count on the device."""

    def __init__(self, depth=101, num_classes=21, image_shape=(800, 1333), num_proposals=1000, dtype=torch.float32,
                 max_batch=1, f32_form='exact', **hot_kwargs):
        super().__init__()
        b = _BLOCKS[depth]
        self.dtype = dtype
        # float32 mode only: 'exact' = exact-float32 matrix instructions (the parity mode), 'x3' = split precision (three
        # bfloat16 limbs per operand, six products per k, float32 accumulation: float32-class accuracy at 2.6 x the peak rate)
        self.f32_form = f32_form
        self.image_shape = (int(image_shape[0]), int(image_shape[1]))
        self.num_classes = num_classes
        # extractor (resnet_fpn.py:228-259, 262-289)
        self.conv1 = _fold_frozen_bn(_conv(3, 64, 7, 2, 0))
        self.conv2 = _stack(64, 64, b[0], 1)
        self.conv3 = _stack(256, 128, b[1], 2)
        self.conv4 = _stack(512, 256, b[2], 2)
        self.conv5 = _stack(1024, 512, b[3], 2)
        # neck (resnet_fpn.py:339-407)
        self.p5 = _conv(2048, 256, 1)
        self.l4, self.l3, self.l2 = _conv(1024, 256, 1), _conv(512, 256, 1), _conv(256, 256, 1)
        self.s4, self.s3, self.s2 = _conv(256, 256, 3, 1, 1), _conv(256, 256, 3, 1, 1), _conv(256, 256, 3, 1, 1)
        # RPN head, shared by the five levels (base_fpn_model.py:393-434); 3 anchors per cell
        self.A = 3
        self.rpn_conv = _conv(256, 512, 3, 1, 1, std=0.01)
        self.rpn_score = _conv(512, 2 * self.A, 1, std=0.01)
        self.rpn_bbox = _conv(512, 4 * self.A, 1, std=0.001)
        # RoI head (resnet_fpn.py:292-336): flatten(7,7,256) -> fc 1024 -> fc 1024 -> score / boxes
        self.fc1 = nn.Linear(7 * 7 * 256, 1024)
        self.fc2 = nn.Linear(1024, 1024)
        self.score = nn.Linear(1024, num_classes)
        self.bbox = nn.Linear(1024, 4 * num_classes)
        for m, std in ((self.fc1, 0.01), (self.fc2, 0.01), (self.score, 0.01), (self.bbox, 0.001)):
            nn.init.normal_(m.weight, 0.0, std)
            nn.init.zeros_(m.bias)
        self._hot_args = (self.image_shape, num_classes, num_proposals, 256)
        self._hot_kwargs = dict(blind_chunks=DEFAULT_BLIND_CHUNKS)
        self._hot_kwargs.update(hot_kwargs)
        self._hot = []
        self._rpn_pair = None
        self._max_batch = max_batch

    def prepare(self, device='cuda'):
        """Moves the model to the GPU and allocates the hot path.  The images of a batch go through the hot path
        in the SAME kernel launches and through the RoI head as one batch (FpnStepBatch; sync-free NMS with
        `blind_chunks` chunks: the first one shared by the batch, the others per image); `batched=False` in
        the hot-path keywords selects the per-image path (FpnHotPath per image)."""
        self.to(device=device, dtype=self.dtype, memory_format=torch.channels_last).eval()
        ops.invalidate_planes(self)                   # (cached limb planes of weights that may have been rewritten through .data)
        fd = torch.float16 if self.dtype == torch.float16 else torch.float32
        self._rpn_pair = None
        self._steps = None
        if self._max_batch <= 64 and self._hot_kwargs.pop('batched', True):
            self._steps = FpnStepBatch(self._max_batch, *self._hot_args, feature_dtype=fd, **self._hot_kwargs)
            self._hot = self._steps.slots
            K = self._hot_args[2]
            dev = next(self.parameters()).device
            self._cls = torch.zeros((self._max_batch, K, self.num_classes), dtype=torch.float32, device=dev)
            self._dlt = torch.zeros((self._max_batch, K, 4 * self.num_classes), dtype=torch.float32, device=dev)
            self._bound = False
        else:
            self._hot = [FpnHotPath(*self._hot_args, feature_dtype=fd, **self._hot_kwargs)
                         for _ in range(self._max_batch)]
        return self

    # ---- dense parts ---------------------------------------------------------------------------
    @_in_f32_form
    def extractor(self, images_nhwc):
        """[B,H,W,3] -> (C2, C3, C4, C5) channels_last (get_resnet_v1_extractor, resnet_fpn.py:262-289)."""
        # conv1_pad + valid 7x7/2, bias + ReLU, pool1_pad (zeros) + 3x3/2 -- float16: one launch from the image; otherwise
        # the last three in one pass (x >= 0 after the ReLU, so skipping the window taps outside the map gives the same
        # maxima as the zero padding)
        x = _stem(self.conv1, images_nhwc, self.dtype)
        c2 = self.conv2(x)
        c3 = self.conv3(c2)
        c4 = self.conv4(c3)
        return c2, c3, c4, self.conv5(c4)

    def features(self, images_nhwc):
        """[B,H,W,3] -> (P2..P6), each [B,256,h,w] channels_last (= NHWC in memory)."""
        return self.neck(self.extractor(images_nhwc))

    @_in_f32_form
    def neck(self, c_list):
        """(C2..C5) -> (P2..P6) (ResnetFpnNeck.call, resnet_fpn.py:378-407)."""
        c2, c3, c4, c5 = c_list
        p5 = _conv_epi(self.p5, c5)
        p6 = p5[:, :, ::2, ::2]                                                  # MaxPooling2D(1x1, stride 2)
        p4 = self._lateral_merge(p5, self.l4, c4)
        p3 = self._lateral_merge(p4, self.l3, c3)
        p2 = self._lateral_merge(p3, self.l2, c2)
        return _conv_epi(self.s2, p2), _conv_epi(self.s3, p3), _conv_epi(self.s4, p4), p5, p6

    def _lateral_merge(self, top, conv, c):
        """P_k = 0.5 * resize_bilinear(P_{k+1}) + 0.5 * lateral(C_k) (resnet_fpn.py:385-398) in ONE launch: the merge rides in
        the epilogue of the lateral 1x1 convolution (ops.lateral_merge: the lateral map is never written; 226 vs 301 us for
        P2 at batch 8)."""
        if not _pw_ok(conv, c) or tuple(conv.stride) != (1, 1) or top.dtype != c.dtype:
            raise _no_kernel('lateral convolution + top-down merge', conv, c)
        return ops.lateral_merge(_nhwc(c), conv.weight, conv.bias, _nhwc(top)).permute(0, 3, 1, 2)

    @staticmethod
    def _merge(top, lateral):
        """0.5 * resize_bilinear(top) + 0.5 * lateral (resnet_fpn.py:385-398) as a launch of its own (ops.fpn_topdown_merge,
        float32 bit-identical to the TF1 restatement): for callers that already hold the lateral map."""
        return ops.fpn_topdown_merge(_nhwc(top), _nhwc(lateral)).permute(0, 3, 1, 2)

    @_in_f32_form
    def rpn(self, p_list):
        """shared RpnHead on every level; outputs concatenated P2->P6 in (y, x, anchor) order
        (base_fpn_model.py:188-200, 427-432): scores [B, N, 2], deltas [B, N, 4] (float32)."""
        p0 = p_list[0]
        if not _own_conv3x3(self.rpn_conv, p0) or self.rpn_conv.out_channels % 256:
            raise _no_kernel('RpnHead', self.rpn_conv, p0)
        # the two 1x1 convolutions run as ONE contraction (weights concatenated: the 512-channel activation is read once)
        w, b = self._rpn_pair_weights()
        B = p0.shape[0]
        n = sum(int(p.shape[2]) * int(p.shape[3]) for p in p_list) * self.A
        scores = torch.empty((B, n, 2), dtype=torch.float32, device=p0.device)
        deltas = torch.empty((B, n, 4), dtype=torch.float32, device=p0.device)
        xs = [_nhwc(p) for p in p_list]
        if p0.dtype == torch.float16:
            if 6 * self.A > 32 or self.rpn_conv.out_channels > 512:
                raise _no_kernel('RpnHead (more than 5 anchors per cell or more than 512 channels)', self.rpn_conv, p0)
            # the WHOLE head in one launch: the 3x3 convolution of all levels with bias + ReLU + both 1x1 convolutions in its
            # epilogue (ops.rpn_head_fused): the 512-channel activation is never written
            return ops.rpn_head_fused(xs, self.rpn_conv.weight, self.rpn_conv.bias, w, b, self.A, scores, deltas)
        # float32 (the parity mode): the 3x3 convolution of all levels in one launch on exact-float32 matrix instructions,
        # bias + ReLU in its epilogue (ops.conv3x3_f32_levels: the small levels' workgroups fill the tail of the big ones'),
        # the two 1x1 convolutions as the exact-float32 GEMM (weight rows zero-padded to 64), then ONE pass per level adds
        # the bias and writes the level's slices of the concatenated arrays (ops.rpn_pack_pair)
        heads = ops.conv3x3_f32_levels(xs, self.rpn_conv.weight, self.rpn_conv.bias, relu=True)
        off = 0
        for h in heads:
            sd = ops.pointwise(h, self._rpn_pair_padded(w), None)[..., :6 * self.A]
            ops.rpn_pack_pair(sd if sd.is_contiguous() else sd.contiguous(), b, self.A, scores, deltas, off)
            off += int(h.shape[1]) * int(h.shape[2]) * self.A
        return scores, deltas

    def _rpn_pair_weights(self):
        return rpn_pair_weights(self)

    def _rpn_pair_padded(self, w):
        return rpn_pair_padded(self, w)

    def head_activation(self, roi_features):
        """flatten(7,7,256) -> fc 1024 -> fc 1024 (resnet_fpn.py:292-326): the input of the score / bbox layers; the Dense
        layers on the pointwise GEMM kernel with bias + ReLU in its epilogue"""
        x = roi_features.reshape(roi_features.shape[0], -1).to(self.dtype)
        x = ops.dense(x if x.is_contiguous() else x.contiguous(), self.fc1.weight, self.fc1.bias, relu=True)
        return ops.dense(x, self.fc2.weight, self.fc2.bias, relu=True)

    @_in_f32_form
    def roi_head(self, roi_features):
        """RoI features [R,7,7,256] -> (class logits [R,Ccls], box regressions [R,4 Ccls]), float32 (resnet_fpn.py:292-336)"""
        return self._final_outputs(self.head_activation(roi_features))

    # ---- HIP-graph replay ---------------------------------------------------------------------------
    def capture(self, batch, warmup=3):
        """Captures forward() for `batch` images of self.image_shape into ONE HIP graph (the whole detector:
        convolutions, neck merges, the sync-free hot path, the RoI head) and
        returns `run(images_nhwc) -> outputs`: the images are copied into the graph's static input and the
        graph is replayed -- a few hundred launches cost one host call, which is what a batch-1 latency
        step is bound by.  Needs the sync-free proposal stage (blind_chunks >= 1: no host check inside).
        A replay makes NO after-pass check (no host sync): `run.recover()` re-runs images whose sync-free NMS did not complete,
        and with f32_form = 'x2' the caller reads `run.range_ok()` (the instance's status word: False = some replay since the
        last call met an activation outside float16's range -- run those images through forward() again)."""
        if not self._hot:
            raise RuntimeError('prepare() first')
        dev = next(self.parameters()).device
        static_in = torch.zeros((batch,) + self.image_shape + (3,), dtype=torch.float32, device=dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):                  # every lazy allocation (weight packs, workspaces)
                self.forward(static_in)
            side.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                static_out = self.forward(static_in)
        torch.cuda.current_stream(dev).wait_stream(side)

        def run(images_nhwc):
            static_in.copy_(images_nhwc)
            graph.replay()
            return static_out

        run.graph = graph
        run.range_ok = self.range_ok
        run.recover = lambda: self.recover(batch)
        return run

    # ---- the model ----------------------------------------------------------------------------------
    def _dense(self, images_nhwc):
        """extractor -> neck -> RPN head: (rpn scores [B,N,2], rpn deltas [B,N,4], NHWC views of P2..P5)"""
        p_list = self.features(images_nhwc)
        rpn_scores, rpn_deltas = self.rpn(p_list)
        rpn_scores, rpn_deltas = rpn_scores.float().contiguous(), rpn_deltas.float().contiguous()
        # float16 maps go to the RoI kernel as they are, anything else as float32
        if self.dtype == torch.float16:
            maps = [p.permute(0, 2, 3, 1) for p in p_list[:4]]
        else:
            maps = [p.permute(0, 2, 3, 1).float() for p in p_list[:4]]
        return rpn_scores, rpn_deltas, maps

    def _hot_to_head(self, B, rpn_scores, rpn_deltas, maps):
        """proposals -> level assignment -> RoI features -> RoI head.  Per image (class softmax [K,Ccls], raw deltas
        [K,4*Ccls]) for the level-sorted RoIs of its hot-path slot; rows >= the image's proposal count are padding."""
        if self._steps is not None:
            # B images in the same hot-path launches, the RoI head on all B x K crops at once
            sb = self._steps
            maps = [m if m.is_contiguous() else m.contiguous() for m in maps]
            bind = sb.rebind if self._bound else sb.bind
            for b in range(B):
                bind(b, rpn_scores[b], rpn_deltas[b], [m[b:b + 1] for m in maps], self._cls[b], self._dlt[b])
            if B == self._max_batch:
                self._bound = True                      # every descriptor has been filled once
            sb.enqueue(sb.STAGE_PROPOSALS | sb.STAGE_ROI, B)
            K = self._cls.shape[1]
            feats = sb.roi_features[:B].reshape((B * K,) + tuple(sb.roi_features.shape[2:]))
            logits, bbox = self.roi_head(feats)
            torch.softmax(logits.float(), dim=-1, out=self._cls[:B].view(B * K, -1))
            self._dlt[:B].view(B * K, -1).copy_(bbox)
            return [(self._cls[b], self._dlt[b]) for b in range(B)]
        heads = []
        for b in range(B):
            hot = self._hot[b]
            hot.stage_proposals(rpn_scores[b], rpn_deltas[b])
            feats = hot.stage_roi([m[b:b + 1].contiguous() for m in maps])
            logits, bbox = self.roi_head(feats)
            heads.append((torch.softmax(logits.float(), dim=-1).contiguous(), bbox.float().contiguous()))
        return heads

    def _run_to_head(self, images_nhwc):
        """base_fpn_model.py:208-265 / :372-382: everything of the inference pass before post_ops_prediction"""
        B = images_nhwc.shape[0]
        if B > len(self._hot):
            raise ValueError('batch %d exceeds max_batch %d' % (B, len(self._hot)))
        rpn_scores, rpn_deltas, maps = self._dense(images_nhwc)
        heads = self._hot_to_head(B, rpn_scores, rpn_deltas, maps)
        self._last_pass = (rpn_scores, rpn_deltas, maps, heads)
        return heads

    @staticmethod
    def _maps_of(maps, b):
        return [m[b:b + 1].contiguous() for m in maps]

    def _detect(self, heads):
        B = len(heads)
        if self._steps is not None:
            sb = self._steps
            sb.enqueue(sb.STAGE_DETECT, B)
            return [(h.det_boxes, h.det_labels, h.det_scores, h.det_count) for h in sb.slots[:B]]
        return [self._hot[b].stage_detect(cls, dlt) for b, (cls, dlt) in enumerate(heads)]

    def _forward_batched(self, B, rpn_scores, rpn_deltas, maps):
        """the hot path + RoI head + post-ops of B images given the dense parts' outputs"""
        return self._detect(self._hot_to_head(B, rpn_scores, rpn_deltas, maps))

    @torch.no_grad()
    def forward(self, images_nhwc, check=None):
        """-> per image (boxes [M,4], labels [M], scores [M], count) padded to max_per_image, count on the device
        (post_ops_prediction, base_fpn_model.py:267-275).  check: see _NmsCompleteness."""
        return self._forward_checked(images_nhwc, check, lambda im: self._detect(self._run_to_head(im)))

    @torch.no_grad()
    def im_detect(self, images_nhwc, img_scale):
        """The evaluation entry of the reference models (base_fpn_model.py:364-390): per image
        (softmax scores [R,Ccls], raw deltas [R,4*Ccls], rois / img_scale [R,4]) for the R proposals the image kept,
        in level-sorted order with empty levels dropped (:384-388) -- what evaluation.pascal_eval.detect_image
        (pascal_eval_files_utils.py:76-106) consumes with img_scale = 1.  img_scale: one number or one per image.
        Host-syncs once (R is data dependent, as in the reference)."""
        heads = self.run_range_checked(lambda: self._run_to_head(images_nhwc))     # ('x2': out of range -> again on three limbs)
        B = len(heads)
        self._last_batch = B
        self.recover(B)
        out = []
        for b, (cls, dlt) in enumerate(heads):
            hot = self._hot[b]
            k = int(hot.roi_count.item())
            sc = img_scale[b] if isinstance(img_scale, (list, tuple)) or (hasattr(img_scale, 'ndim') and img_scale.ndim > 0) else img_scale
            # tensor / tensor: a true float32 division per element (tensor / python-number multiplies by the reciprocal
            # on the GPU, which is not what tf.to_float(img_scale) division gives)
            div = torch.full((1,), float(sc), dtype=torch.float32, device=hot.sorted_rois.device)
            out.append((cls[:k].clone(), dlt[:k].clone(), hot.sorted_rois[:k] / div))
        return out

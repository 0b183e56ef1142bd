"""ResNet-FPN detector assembled around the HIP hot path -- counterpart of the reference's
model/fpn/resnet_fpn.py (ResnetV1Fpn: extractor, ResnetFpnNeck, ResnetRoiHead) + the inference
branch of model/fpn/base_fpn_model.py (BaseFPN.call, RpnHead).

SURVEY.md section 8(f) ranks 2-3 ("next"): the dense conv / FC stacks are genuine dense contractions
and run on the MFMA units through PyTorch-ROCm's library convolutions (MIOpen / hipBLASLt) in NHWC
(channels_last), fp32 or fp16; everything between them is the hand-written hot path (FpnHotPath).
Weights are randomly initialised with the reference's initialisers (no checkpoints exist offline);
frozen batch-norm (epsilon 1.001e-5, inference statistics) is folded into the convolutions.

Shapes follow the reference exactly so that feature maps and anchor grids agree:
conv1 = pad 3 + 7x7/2 valid, pool1 = pad 1 + 3x3/2 valid, the stride of a stage sits on the first
1x1 convolution of its first block, P6 = P5[::2, ::2], top-down merge = 0.5 * resize_bilinear(P_{k+1})
+ 0.5 * lateral with TF1's legacy resize (src = dst * in/out, no half-pixel offset) -- one fused HIP
kernel per merge on the GPU (ops.fpn_topdown_merge).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

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


# 1x1 stride-1 convolutions in NHWC are plain GEMMs on the [pixels, channels] view and most of them are bound by HBM
# traffic, not by MFMA throughput (tools/exp/conv1x1_gemm.py).  Three routes, measured once per layer shape:
#   'conv' library convolution without bias + the fused epilogue pass (ops.bias_act_)
#   'gemm' hipBLASLt GEMM with bias (+ ReLU) in its own epilogue (no shortcut)
#   'mfma' ops.conv1x1_f16: the hand-written MFMA kernel with bias + shortcut + ReLU fused (float16, cin <= 256)
#   'pw'   ops.pointwise_f16: the LDS-staged GEMM form of the implicit-GEMM kernel (float16, cin >= 128; any stride)
_GEMM_ROUTE = {}
_ROUTE_MODE = __import__('os').environ.get('ODET_ROUTE_1X1', 'table')
# ODET_CONV3X3=lib: the library convolution instead of the hand-written implicit GEMM (ops.conv3x3_f16) where it applies;
# twopass: the hand-written convolution, but the RpnHead as convolution launch + tail launches (not ops.rpn_head_fused)
_CONV3X3_MODE = __import__('os').environ.get('ODET_CONV3X3', 'own')


def _gemm_1x1(conv, x, bias, relu):
    B, cin, h, w = x.shape
    x2 = x.permute(0, 2, 3, 1).reshape(-1, cin)
    w2 = conv.weight.view(conv.out_channels, cin).t()
    y2 = torch._addmm_activation(bias, x2, w2, use_gelu=False) if relu else torch.addmm(bias, x2, w2)
    return y2.view(B, h, w, conv.out_channels).permute(0, 3, 1, 2)


def _conv_1x1(conv, x, bias, relu, res):
    y = F.conv2d(x, conv.weight, None)
    if not y.is_contiguous(memory_format=torch.channels_last):
        y = y.contiguous(memory_format=torch.channels_last)
    ops.bias_act_(y.permute(0, 2, 3, 1), bias, res, relu)
    return y


def _mfma_1x1(conv, x, bias, relu, res, in_bias=None):
    return ops.conv1x1_f16(x.permute(0, 2, 3, 1), conv.weight, bias, res, relu, in_bias=in_bias).permute(0, 3, 1, 2)


# ODET_PW=0: the layers of the pointwise GEMM kernel back on their round-2 routes (library GEMM / convolution); a comma list of
# names from {c1, s2, short, lateral, fc, final} switches single classes of layers off (same-box A/B runs)
_PW_OFF = set(filter(None, __import__('os').environ.get('ODET_PW_OFF', '').split(',')))
if __import__('os').environ.get('ODET_PW', '1') == '0':
    _PW_OFF = {'c1', 's2', 'short', 'lateral', 'fc', 'final'}


def _pw_ok(conv, x):
    """the pointwise GEMM kernel takes this 1x1 convolution (stride 1 or 2, no padding)"""
    if not x.is_cuda or x.dtype not in (torch.float16, torch.float32) or ('f32' in _PW_OFF and x.dtype == torch.float32):
        return False
    gran = 64 if x.dtype == torch.float16 else 32          # channels per K-step
    return (tuple(conv.kernel_size) == (1, 1) and tuple(conv.padding) == (0, 0)
            and tuple(conv.stride) in ((1, 1), (2, 2)) and conv.in_channels % gran == 0 and conv.in_channels >= 2 * gran
            and conv.out_channels % 64 == 0 and x.is_contiguous(memory_format=torch.channels_last))


def _pw_1x1(conv, x, bias, relu, res):
    return ops.pointwise(x.permute(0, 2, 3, 1), conv.weight, bias, res, relu, conv.stride[0]).permute(0, 3, 1, 2)


def _time_route(fn, reps=5):
    for _ in range(2):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b)


def _route_1x1(conv, x, bias, relu, res):
    """'conv', 'gemm' or 'mfma' for this layer shape.  Deterministic (the routes differ in rounding: the MFMA kernel
    adds shortcut + W.x + bias with one rounding): the table below is what timing the three routes chose on MI355X for
    every 1x1 layer shape of the three detectors (tools/exp/conv1x1_routes.py) -- float16 layers with a shortcut and
    the small-K layers without one go to the hand-written MFMA kernel, layers without a shortcut otherwise to the GEMM
    with its own epilogue, the rest to the library convolution + epilogue pass.  ODET_ROUTE_1X1=measure re-times the
    routes at first use (per device and shape), ODET_ROUTE_1X1=conv|gemm|mfma forces one where it applies."""
    mfma_ok = x.dtype == torch.float16 and conv.in_channels in (64, 128, 256, 512) and conv.out_channels % 64 == 0
    mode = _ROUTE_MODE
    if mode == 'measure':
        key = (x.device.index, tuple(x.shape), conv.out_channels, x.dtype, bool(relu), res is not None)
        r = _GEMM_ROUTE.get(key)
        if r is None:
            if torch.cuda.is_current_stream_capturing():
                mode = 'table'
            else:
                cand = {'conv': lambda: _conv_1x1(conv, x, bias, relu, res)}
                if res is None:
                    cand['gemm'] = lambda: _gemm_1x1(conv, x, bias, relu)
                if mfma_ok:
                    cand['mfma'] = lambda: _mfma_1x1(conv, x, bias, relu, res)
                if _pw_ok(conv, x):
                    cand['pw'] = lambda: _pw_1x1(conv, x, bias, relu, res)
                times = {k: _time_route(f) for k, f in cand.items()}
                r = min(times, key=times.get)
                _GEMM_ROUTE[key] = r
                return r
        else:
            return r
    if mode == 'mfma' and mfma_ok:
        return 'mfma'
    if mode == 'gemm' and res is None:
        return 'gemm'
    if mode == 'conv':
        return 'conv'
    pw_ok = _pw_ok(conv, x) and 'c1' not in _PW_OFF
    # (tools/exp/pointwise_layers.py, pointwise_tiles.py: the register-resident kernel keeps the layers with a shortcut up
    # to 256 input channels and the 64-channel outputs; everything else -- every first 1x1 of a bottleneck, K = 512 with a
    # shortcut (38 vs 49 us), the neck's P5 -- is ahead or level on the LDS-staged GEMM)
    if mfma_ok and conv.in_channels <= 256 and (res is not None or conv.out_channels <= 64 or not pw_ok):
        return 'mfma'
    if pw_ok:
        return 'pw'
    if mfma_ok:
        return 'mfma'
    if res is None:
        return 'gemm'
    return 'conv'


def _own_conv3x3(conv, x, pad=None):
    """True when the hand-written implicit-GEMM kernel (ops.conv3x3_f16) takes this 3x3 convolution: float16 NHWC,
    stride 1, padding 1, cin % 64 == 0, cout % 128 == 0, and enough output tiles to fill the chip (the kernel cuts the
    pixels into 128 .. 256-row slabs; measured on the detectors' layer shapes at batch 1 / 4 / 8,
    tools/exp/conv3x3_layers.py: 1.3-1.9x ahead of the library from ~100 workgroups of the smallest tile on, behind it
    on the small maps)."""
    if _CONV3X3_MODE == 'lib' or x.dtype not in (torch.float16, torch.float32) or pad is not None or not x.is_cuda:
        return False
    if tuple(conv.kernel_size) != (3, 3) or tuple(conv.stride) != (1, 1) or tuple(conv.padding) != (1, 1):
        return False
    if x.dtype == torch.float32:
        # the parity mode: exact-float32 matrix instructions, always the own kernel (ops.conv3x3_f32; on a par with the
        # library's float32 rate, and no solver search whose choice could change the summation order between runs)
        return ('f32' not in _PW_OFF and conv.in_channels % 32 == 0 and conv.out_channels % 64 == 0
                and x.is_contiguous(memory_format=torch.channels_last))
    # (ResNet conv2's 64 -> 64 layer on 64-channel tiles: 82 vs 87 us at batch 8, 138 vs 160 at 15, since the LDS stages are
    # sized by the tile and several of the small workgroups share a CU)
    if conv.in_channels % 64 != 0 or conv.out_channels % 64 != 0:
        return False
    if not x.is_contiguous(memory_format=torch.channels_last):
        return False
    m = int(x.shape[0]) * int(x.shape[2]) * int(x.shape[3])
    co = conv.out_channels
    tiles_n = co // 256 if co % 256 == 0 else (co // 128 if co % 128 == 0 else co // 64)
    if 'c3x3_64' in _PW_OFF and co % 128 != 0:
        return False
    return _CONV3X3_MODE == 'force' or ((m + 127) // 128) * tiles_n >= 100


# the float32 mode's patch matrices (stem, VGG16's first convolution) are addressed with 32-bit byte offsets
_PATCH_BYTES_MAX = 0xF0000000


def _stem(conv1, images_nhwc, dtype):
    """conv1_pad + 7x7/2 'valid' + folded BN + ReLU + pool1_pad + 3x3/2 max-pooling (resnet_fpn.py:262-289).  float16 on the
    GPU: ONE launch from the image (ops.stem_conv7_pool3: the 64-channel convolution output, 273 MB at batch 8, never
    goes to memory); otherwise the library convolution + the fused bias / ReLU / pooling pass."""
    if (_CONV3X3_MODE in ('own', 'force') and images_nhwc.is_cuda and dtype == torch.float16 and conv1.out_channels == 64
            and images_nhwc.dtype in (torch.float32, torch.float16) and images_nhwc.is_contiguous()):
        key = (conv1.weight.data_ptr(), conv1.weight._version)
        packed = getattr(conv1, '_odet_packed', None)
        if packed is None or packed[0] != key:
            packed = (key, ops.stem_pack_weights(conv1.weight))
            conv1._odet_packed = packed
        return ops.stem_conv7_pool3(images_nhwc, packed[1], conv1.bias).permute(0, 3, 1, 2)
    if (images_nhwc.is_cuda and dtype == torch.float32 and images_nhwc.dtype == torch.float32 and images_nhwc.is_contiguous()
            and conv1.out_channels == 64 and tuple(conv1.kernel_size) == (7, 7) and 'f32' not in _PW_OFF
            and _CONV3X3_MODE in ('own', 'force')):
        # float32 (parity mode): the 7x7 / 2 convolution as the exact-float32 GEMM on its patch matrix (ops.stem_patches_f32:
        # 160 floats per output pixel), then bias + ReLU + the 3x3 / 2 pooling in one pass -- no library convolution
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
    x = images_nhwc.to(dtype).permute(0, 3, 1, 2)                               # NHWC memory, NCHW view
    return _conv_relu_pool(conv1, x, 3, 2, pool_pad=1, pad=(3, 3, 3, 3))


def _conv_relu_pool(conv, x, kernel, stride, pool_pad=0, ceil_mode=False, pad=None):
    """max_pool(relu(conv(x) + bias)): on the GPU the convolution runs without its bias and ONE pass
    (ops.bias_relu_maxpool) reads its output once and writes the pooled map; torch formulation elsewhere."""
    if x.is_cuda and x.dtype in (torch.float32, torch.float16) and conv.out_channels % 8 == 0:
        if (_own_conv3x3(conv, x, pad) and x.dtype == torch.float16 and kernel == 2 and stride == 2 and pool_pad == 0 and ceil_mode
                and conv.bias is not None and 'pool' not in _PW_OFF):
            # Conv2D + ReLU + MaxPooling2D((2, 2), 2, 'same') in the convolution's own launch (its pixel order makes a pooling
            # window four neighbouring lanes): the un-pooled map is never written
            return ops.conv3x3_relu_pool2_f16(x.permute(0, 2, 3, 1), conv.weight, conv.bias).permute(0, 3, 1, 2)
        if _own_conv3x3(conv, x, pad):
            # the hand-written implicit GEMM without its epilogue, then the fused bias + ReLU + pooling pass
            yn = (ops.conv3x3_f32 if x.dtype == torch.float32 else ops.conv3x3_f16)(x.permute(0, 2, 3, 1), conv.weight)
            return ops.bias_relu_maxpool(yn, conv.bias, kernel, stride, pool_pad, ceil_mode).permute(0, 3, 1, 2)
        padding = conv.padding
        if pad is not None:
            if pad[0] == pad[1] == pad[2] == pad[3] and tuple(conv.padding) == (0, 0):
                padding = (pad[0], pad[0])
            else:
                x = F.pad(x, pad)
        y = F.conv2d(x, conv.weight, None, conv.stride, padding)
        if not y.is_contiguous(memory_format=torch.channels_last):
            y = y.contiguous(memory_format=torch.channels_last)
        return ops.bias_relu_maxpool(y.permute(0, 2, 3, 1), conv.bias, kernel, stride, pool_pad, ceil_mode).permute(0, 3, 1, 2)
    y = _conv_epi(conv, x, relu=True, pad=pad)
    return F.max_pool2d(y, kernel, stride, padding=pool_pad, ceil_mode=ceil_mode)


def _conv_epi(conv, x, relu=False, residual=None, extra_bias=None, pad=None):
    """conv (+ zero padding `pad`) -> + bias (+ extra_bias) (+ residual) -> ReLU.  On the GPU the convolution
    runs without its bias and everything after it is ONE in-place pass of the fused HIP epilogue
    (ops.bias_act_) over the NHWC output; the torch formulation serves the CPU shape-bookkeeping test."""
    padding = conv.padding
    if pad is not None:
        if pad[0] == pad[1] == pad[2] == pad[3] and tuple(conv.padding) == (0, 0):
            padding = (pad[0], pad[0])           # symmetric ZeroPadding2D + 'valid' == the convolution's own zero padding
        else:
            x = F.pad(x, pad)
    bias = conv.bias if extra_bias is None else conv.bias + extra_bias
    if x.is_cuda and x.dtype in (torch.float32, torch.float16) and conv.out_channels % 8 == 0:
        if (pad is None and tuple(conv.kernel_size) == (1, 1) and tuple(conv.stride) == (1, 1)
                and tuple(conv.padding) == (0, 0) and x.is_contiguous(memory_format=torch.channels_last)):
            res = None
            if residual is not None:
                res = residual.permute(0, 2, 3, 1)
                if not res.is_contiguous():
                    res = res.contiguous()
            route = _route_1x1(conv, x, bias, relu, res)
            if route == 'pw':
                return _pw_1x1(conv, x, bias, relu, res)
            if route == 'gemm':
                return _gemm_1x1(conv, x, bias, relu)
            if route == 'mfma':
                return _mfma_1x1(conv, x, bias, relu, res)
            return _conv_1x1(conv, x, bias, relu, res)
        if pad is None and _ROUTE_MODE == 'table' and 's2' not in _PW_OFF and tuple(conv.stride) == (2, 2) and _pw_ok(conv, x):
            # Conv2D(1x1, strides 2, 'valid') = the same GEMM over every second pixel (the first block of a stage)
            res = None
            if residual is not None:
                res = residual.permute(0, 2, 3, 1)
                if not res.is_contiguous():
                    res = res.contiguous()
            return _pw_1x1(conv, x, bias, relu, res)
        if residual is None and _own_conv3x3(conv, x, pad):
            # hand-written implicit GEMM on the matrix cores with bias (+ ReLU) in its epilogue
            if x.dtype == torch.float32:
                return ops.conv3x3_f32(x.permute(0, 2, 3, 1), conv.weight, bias, relu=relu).permute(0, 3, 1, 2)
            b16 = bias if bias.dtype == torch.float16 else bias.half()
            return ops.conv3x3_f16(x.permute(0, 2, 3, 1), conv.weight, b16, relu=relu).permute(0, 3, 1, 2)
        y = F.conv2d(x, conv.weight, None, conv.stride, padding)
        if not y.is_contiguous(memory_format=torch.channels_last):
            y = y.contiguous(memory_format=torch.channels_last)
        res = None
        if residual is not None:
            res = residual.permute(0, 2, 3, 1)
            if not res.is_contiguous():
                res = res.contiguous()
        ops.bias_act_(y.permute(0, 2, 3, 1), bias, res, relu)
        return y
    y = F.conv2d(x, conv.weight, bias, conv.stride, padding)
    if residual is not None:
        y = y + residual
    return F.relu(y) if relu else y


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
        """[w3 | w_shortcut] along K and b3 + b_shortcut (ops.pointwise_dual_f16), cached until a parameter changes"""
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
        # Add([shortcut, x]) + ReLU ride on c3's epilogue; a convolutional shortcut runs without its bias,
        # which is added to c3's instead
        if (self.short is not None and x.is_cuda and x.dtype in (torch.float16, torch.float32) and _ROUTE_MODE == 'table'
                and 'dual' not in _PW_OFF and not ('f32' in _PW_OFF and x.dtype == torch.float32)
                and self.c3.in_channels % 64 == 0 and self.short.in_channels % 64 == 0 and self.c3.out_channels % 64 == 0
                and tuple(self.short.stride) in ((1, 1), (2, 2)) and x.is_contiguous(memory_format=torch.channels_last)):
            # a stage's first block, float16: the last 1x1 convolution AND the convolutional shortcut as ONE contraction
            # over [c2's output | the block's (strided) input] with the weights concatenated along K
            # (ops.pointwise_dual_f16) -- the shortcut map is never written or re-read, one launch instead of two
            y = _conv_epi(self.c2, _conv_epi(self.c1, x, relu=True), relu=True)
            w, b = self._dual_weights()
            return ops.pointwise_dual(y.permute(0, 2, 3, 1), x.permute(0, 2, 3, 1), w, b, self.short.stride[0],
                                      relu=True).permute(0, 3, 1, 2)
        if self.short is None:
            sc, sb = x, None
        elif x.is_cuda and x.dtype == torch.float16 and _ROUTE_MODE == 'table' and 'short' not in _PW_OFF and (
                _pw_ok(self.short, x) or (tuple(self.short.stride) == (1, 1) and self.short.in_channels == 64
                                          and x.is_contiguous(memory_format=torch.channels_last))):
            # float16: the shortcut convolution WITH its bias on the own kernels (strided or long K: the pointwise GEMM;
            # ResNet conv2's 64 -> 256: the register-resident kernel) -- no library convolution, no separate bias add
            sc, sb = _conv_epi(self.short, x, relu=False), None
        else:
            sc, sb = F.conv2d(x, self.short.weight, None, self.short.stride, self.short.padding), self.short.bias
        y = _conv_epi(self.c1, x, relu=True)
        if y.is_cuda and y.dtype == torch.float16 and self.c3.in_channels in (64, 128, 256, 512):
            # c2 (3x3) runs WITHOUT bias / ReLU; if c3 takes the MFMA route for this shape, c2's epilogue is applied
            # to c3's operand fragments as they are loaded (ops.conv1x1_f16(in_bias=...)) and its pass disappears
            if self.c2.out_channels in ((256,) if 'tail' in _PW_OFF else (64, 128, 256)) and _CONV3X3_MODE in ('own', 'force') \
                    and _own_conv3x3(self.c2, y) and (int(y.shape[0]) * int(y.shape[2]) * int(y.shape[3]) + 127) // 128 >= 128:
                # the 3x3 convolution AND the block's last 1x1 convolution + bias + shortcut + ReLU in one launch
                # (ops.conv3x3_conv1x1_f16: the 64 / 128 / 256-channel activation between them stays in LDS): conv4 83 vs 108 us
                # at batch 8 (253 vs 320 at 30), conv3 122 vs 134 (400 vs 477), conv2 198 vs 226 (640-700 vs 787); behind the
                # two launches on smaller maps (tools/exp/block_tail_layers.py)
                res = sc.permute(0, 2, 3, 1)
                if not res.is_contiguous():
                    res = res.contiguous()
                b3 = self.c3.bias if sb is None else self.c3.bias + sb
                out = ops.conv3x3_conv1x1_f16(y.permute(0, 2, 3, 1), self.c2.weight, self.c2.bias, self.c3.weight, b3,
                                              residual=res, relu=True)
                return out.permute(0, 3, 1, 2)
            res = sc.permute(0, 2, 3, 1)
            if not res.is_contiguous():
                res = res.contiguous()
            b3 = self.c3.bias if sb is None else self.c3.bias + sb
            if _route_1x1(self.c3, y, b3, True, res) == 'mfma':           # (c2's output has y's shape)
                if _own_conv3x3(self.c2, y):
                    y2 = ops.conv3x3_f16(y.permute(0, 2, 3, 1), self.c2.weight).permute(0, 3, 1, 2)
                else:
                    y2 = F.conv2d(y, self.c2.weight, None, self.c2.stride, self.c2.padding)
                    if not y2.is_contiguous(memory_format=torch.channels_last):
                        y2 = y2.contiguous(memory_format=torch.channels_last)
                return _mfma_1x1(self.c3, y2, b3, True, res, in_bias=self.c2.bias)
        y = _conv_epi(self.c2, y, relu=True)
        return _conv_epi(self.c3, y, relu=True, residual=sc, extra_bias=sb)


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
        own = (x.is_cuda and x.dtype in (torch.float16, torch.float32) and _ROUTE_MODE == 'table' and x.is_contiguous()
               and 'final' not in _PW_OFF and not ('f32' in _PW_OFF and x.dtype == torch.float32))
        if own:
            wpad, b32 = self._final_layer()
            if wpad is not None:
                y = ops.dense(x, wpad, b32) if x.dtype == torch.float32 else ops.dense_f16_out_f32(x, wpad, b32)
                return y[:, :n1], y[:, n1:n5]
        if x.dtype == torch.float16:
            wpad, b32 = self._final_layer()
            wsrc = torch.cat([self.score.weight, self.bbox.weight], 0) if wpad is None else wpad[:n5]
            y = torch.addmm(b32[:n5], x.float(), wsrc.float().t())
            return y[:, :n1], y[:, n1:n5]
        return self.score(x), self.bbox(x)


class _NmsCompleteness:
    """The detectors run the proposal stage sync-free (no host check between kernels; graph-capturable) with a fixed
    number of NMS chunks.  If an image needs more than those, the hot path reports it EMPTY and flags it
    (nms_done = 0, include/odet.h): `forward()` checks the flags after the last launch of the pass whenever it is not
    being captured into a HIP graph and `check_nms` is on (default), and raises.  Throughput loops that do not want a
    host sync per pass set `model.check_nms = False` and call `check_complete()` themselves (after a graph replay too)."""

    check_nms = True
    _last_batch = 0

    def nms_done(self, batch=None):
        """device int32 flags (1 = complete) of the images of the last pass"""
        n = self._last_batch if batch is None else batch
        return [h.nms_done for h in self._hot[:n]]

    def check_complete(self, batch=None):
        steps = getattr(self, '_steps', None)
        if steps is not None and hasattr(steps, 'nms_done_all'):
            n = self._last_batch if batch is None else batch
            flags = steps.nms_done_all[:n].tolist()                       # one device -> host copy for the whole pass
        else:
            flags = [int(t.item()) for t in self.nms_done(batch)]
        bad = [b for b, f in enumerate(flags) if f != 1]
        if bad:
            raise RuntimeError('the RPN NMS of image(s) %s did not complete inside blind_chunks = %d sync-free chunks: their '
                               'results are reported EMPTY.  Build the detector with more chunks (blind_chunks=...) or a '
                               'wider first chunk (nms_first_chunk=4096)' % (bad, self._hot[0].blind_chunks))

    def _after_pass(self, batch, check):
        self._last_batch = batch
        if check is None:
            check = self.check_nms and not torch.cuda.is_current_stream_capturing()
        if check:
            self.check_complete(batch)


class ResNetFpnDetector(_NmsCompleteness, _FinalLayer, nn.Module):
    """Inference-only ResNet-{50,101,152}-FPN detector.  `forward(images)` takes NHWC float images
    [B,H,W,3] (already mean-subtracted, as the reference's input pipeline delivers them) and returns,
    per image, the padded detections of post_ops_prediction plus their count on the device."""

    def __init__(self, depth=101, num_classes=21, image_shape=(800, 1333), num_proposals=1000, dtype=torch.float32,
                 max_batch=1, **hot_kwargs):
        super().__init__()
        b = _BLOCKS[depth]
        self.dtype = dtype
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
    def features(self, images_nhwc):
        """[B,H,W,3] -> (P2..P6), each [B,256,h,w] channels_last (= NHWC in memory)."""
        # conv1_pad + valid 7x7/2, bias + ReLU, pool1_pad (zeros) + 3x3/2 -- float16: one launch from the image; otherwise
        # the last three in one pass (x >= 0 after the ReLU, so skipping the window taps outside the map gives the same
        # maxima as the zero padding)
        x = _stem(self.conv1, images_nhwc, self.dtype)
        c2 = self.conv2(x)
        c3 = self.conv3(c2)
        c4 = self.conv4(c3)
        c5 = self.conv5(c4)
        p5 = _conv_epi(self.p5, c5)
        p6 = p5[:, :, ::2, ::2]                                                  # MaxPooling2D(1x1, stride 2)
        p4 = self._lateral_merge(p5, self.l4, c4)
        p3 = self._lateral_merge(p4, self.l3, c3)
        p2 = self._lateral_merge(p3, self.l2, c2)
        return _conv_epi(self.s2, p2), _conv_epi(self.s3, p3), _conv_epi(self.s4, p4), p5, p6

    def _lateral_merge(self, top, conv, c):
        """P_k = 0.5 * resize_bilinear(P_{k+1}) + 0.5 * lateral(C_k) (resnet_fpn.py:385-398).  float16: ONE launch -- the
        merge rides in the epilogue of the lateral 1x1 convolution (ops.lateral_merge_f16: the lateral map is never
        written; 226 vs 301 us for P2 at batch 8); otherwise the convolution, then the merge launch."""
        if _ROUTE_MODE == 'table' and 'lateral' not in _PW_OFF and _pw_ok(conv, c) and tuple(conv.stride) == (1, 1) \
                and top.dtype == c.dtype:
            t = top.permute(0, 2, 3, 1)
            t = t if t.is_contiguous() else t.contiguous()
            return ops.lateral_merge(c.permute(0, 2, 3, 1), conv.weight, conv.bias, t).permute(0, 3, 1, 2)
        return self._merge(top, _conv_epi(conv, c))

    @staticmethod
    def _merge(top, lateral):
        """0.5 * resize_bilinear(top) + 0.5 * lateral (resnet_fpn.py:385-398).  On the GPU one launch of the
        fused HIP kernel (odet_fpn_topdown_merge) on the NHWC memory of the channels_last tensors; the torch
        formulation only serves the CPU shape-bookkeeping test and dtypes the kernel does not take."""
        if top.is_cuda and top.dtype in (torch.float32, torch.float16):
            out = ops.fpn_topdown_merge(top.permute(0, 2, 3, 1), lateral.permute(0, 2, 3, 1))
            return out.permute(0, 3, 1, 2)
        return tf_legacy_resize_bilinear(top, lateral.shape[2:]) * 0.5 + lateral * 0.5

    def rpn(self, p_list):
        """shared RpnHead on every level; outputs concatenated P2->P6 in (y, x, anchor) order
        (base_fpn_model.py:188-200, 427-432): scores [B, N, 2], deltas [B, N, 4]."""
        if p_list[0].is_cuda and p_list[0].dtype in (torch.float32, torch.float16):
            # GPU: the two 1x1 convolutions run as ONE contraction (weights concatenated: the 512-channel activation
            # is read once) without bias, and ONE pass per level (ops.rpn_pack_pair) adds the bias, widens to float32
            # and writes the level's slices of the concatenated arrays the proposal stage reads
            w, b = self._rpn_pair_weights()
            B = p_list[0].shape[0]
            n = sum(int(p.shape[2]) * int(p.shape[3]) for p in p_list) * self.A
            scores = torch.empty((B, n, 2), dtype=torch.float32, device=p_list[0].device)
            deltas = torch.empty((B, n, 4), dtype=torch.float32, device=p_list[0].device)
            off = 0
            fused = p_list[0].dtype == torch.float16 and self.rpn_conv.out_channels == 512 and self.A <= 4
            convs = None
            if fused and _CONV3X3_MODE in ('own', 'force') and self.rpn_conv.in_channels % 64 == 0:
                # the WHOLE head in one launch: the 3x3 convolution of all levels with bias + ReLU + both 1x1 convolutions
                # in its epilogue (ops.rpn_head_fused): the 512-channel activation is never written
                xs = [p.permute(0, 2, 3, 1) for p in p_list]
                xs = [x if x.is_contiguous() else x.contiguous() for x in xs]
                return ops.rpn_head_fused(xs, self.rpn_conv.weight, self.rpn_conv.bias, w, b, self.A, scores, deltas)
            if fused and _CONV3X3_MODE != 'lib' and self.rpn_conv.in_channels % 64 == 0:
                # the 3x3 convolution of ALL levels in one launch of the hand-written implicit-GEMM kernel
                # (ops.conv3x3_f16_levels: the small levels' workgroups fill the tail of the big ones'), without bias
                xs = [p.permute(0, 2, 3, 1) for p in p_list]
                xs = [x if x.is_contiguous() else x.contiguous() for x in xs]
                convs = ops.conv3x3_f16_levels(xs, self.rpn_conv.weight)
            heads = None
            if not fused and p_list[0].dtype == torch.float32 and _CONV3X3_MODE != 'lib' \
                    and self.rpn_conv.in_channels % 32 == 0 and self.rpn_conv.out_channels % 256 == 0:
                # float32 (the parity mode): the same grouped launch on exact-float32 matrix instructions, bias + ReLU
                # in its epilogue (ops.conv3x3_f32_levels; on a par with the library on P2 alone, ahead of its five
                # separate launches because the small levels fill the tail: tools/exp/conv3x3_layers.py 8 f32)
                xs = [p.permute(0, 2, 3, 1) for p in p_list]
                xs = [x if x.is_contiguous() else x.contiguous() for x in xs]
                heads = ops.conv3x3_f32_levels(xs, self.rpn_conv.weight, self.rpn_conv.bias, relu=True)
            for li, p in enumerate(p_list):
                if fused:
                    # float16: the 3x3 convolution without bias, then ONE MFMA pass does bias + ReLU + both 1x1
                    # convolutions + their biases + the float32 re-layout (ops.rpn_head_tail)
                    if convs is not None:
                        ops.rpn_head_tail(convs[li], self.rpn_conv.bias, w, b, self.A, scores, deltas, off)
                        off += int(p.shape[2]) * int(p.shape[3]) * self.A
                        continue
                    c = F.conv2d(p, self.rpn_conv.weight, None, 1, self.rpn_conv.padding)
                    if not c.is_contiguous(memory_format=torch.channels_last):
                        c = c.contiguous(memory_format=torch.channels_last)
                    ops.rpn_head_tail(c.permute(0, 2, 3, 1), self.rpn_conv.bias, w, b, self.A, scores, deltas, off)
                else:
                    x = heads[li].permute(0, 3, 1, 2) if heads is not None else _conv_epi(self.rpn_conv, p, relu=True)
                    if heads is not None and 'f32' not in _PW_OFF and self.rpn_conv.out_channels % 32 == 0:
                        # float32: the two 1x1 convolutions as the exact-float32 GEMM (weight rows zero-padded to 64)
                        sd = ops.pointwise(heads[li], self._rpn_pair_padded(w), None)[..., :6 * self.A]
                    else:
                        sd = F.conv2d(x, w, None).permute(0, 2, 3, 1)
                    ops.rpn_pack_pair(sd if sd.is_contiguous() else sd.contiguous(), b, self.A, scores, deltas, off)
                off += int(p.shape[2]) * int(p.shape[3]) * self.A
            return scores, deltas
        scores, deltas = [], []
        for p in p_list:
            x = _conv_epi(self.rpn_conv, p, relu=True)
            B = x.shape[0]
            scores.append(self.rpn_score(x).permute(0, 2, 3, 1).reshape(B, -1, 2))
            deltas.append(self.rpn_bbox(x).permute(0, 2, 3, 1).reshape(B, -1, 4))
        return torch.cat(scores, 1), torch.cat(deltas, 1)

    def _rpn_pair_weights(self):
        return rpn_pair_weights(self)

    def _rpn_pair_padded(self, w):
        return rpn_pair_padded(self, w)

    def roi_head(self, roi_features):
        x = roi_features.reshape(roi_features.shape[0], -1).to(self.dtype)
        own = (x.is_cuda and x.dtype in (torch.float16, torch.float32) and _ROUTE_MODE == 'table' and x.is_contiguous()
               and not ('f32' in _PW_OFF and x.dtype == torch.float32))
        if own and 'fc' not in _PW_OFF:
            # the Dense layers on the pointwise GEMM kernel with bias + ReLU in its epilogue (resnet_fpn.py:292-336)
            x = ops.dense(x, self.fc1.weight, self.fc1.bias, relu=True)
            x = ops.dense(x, self.fc2.weight, self.fc2.bias, relu=True)
        else:
            x = F.relu(self.fc1(x))
            x = F.relu(self.fc2(x))
        return self._final_outputs(x)

    # ---- HIP-graph replay ---------------------------------------------------------------------------
    def capture(self, batch, warmup=3):
        """Captures forward() for `batch` images of self.image_shape into ONE HIP graph (the whole detector:
        library convolutions, fused epilogues / neck merges, the sync-free hot path, the RoI head) and
        returns `run(images_nhwc) -> outputs`: the images are copied into the graph's static input and the
        graph is replayed -- a few hundred launches cost one host call, which is what a batch-1 latency
        step is bound by.  Needs the sync-free proposal stage (blind_chunks >= 1: no host check inside)."""
        if not self._hot:
            raise RuntimeError('prepare() first')
        dev = next(self.parameters()).device
        static_in = torch.zeros((batch,) + self.image_shape + (3,), dtype=torch.float32, device=dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):                  # MIOpen solver search + every lazy allocation
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
        return self._hot_to_head(B, *self._dense(images_nhwc))

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
        heads = self._run_to_head(images_nhwc)
        B = len(heads)
        outs = self._detect(heads)
        self._after_pass(B, check)
        return outs

    @torch.no_grad()
    def im_detect(self, images_nhwc, img_scale):
        """The evaluation entry of the reference models (base_fpn_model.py:364-390): per image
        (softmax scores [R,Ccls], raw deltas [R,4*Ccls], rois / img_scale [R,4]) for the R proposals the image kept,
        in level-sorted order with empty levels dropped (:384-388) -- what evaluation.pascal_eval.detect_image
        (pascal_eval_files_utils.py:76-106) consumes with img_scale = 1.  img_scale: one number or one per image.
        Host-syncs once (R is data dependent, as in the reference)."""
        heads = self._run_to_head(images_nhwc)
        B = len(heads)
        self._last_batch = B
        self.check_complete(B)
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

// Convolution epilogue of the dense path ("frozen-BN folded into conv"): what follows every
// library convolution of the reference's ResNet / neck / RPN head, in ONE in-place pass over the NHWC output:
//     y = conv + bias                        (frozen BatchNormalization folded into weight and bias)
//     y = y + shortcut                       (bottleneck blocks, resnet_fpn.py:154-205: Add([shortcut, x]))
//     y = max(y, 0)                          (Activation('relu'))
// As separate framework ops this is 2-3 launches and 5-7 passes over the activation per convolution; the
// detector spent more time in them than in the convolutions themselves.  HBM-bound: 16 B per lane, bias (C
// values) from L1.  float32: the same operation order as the separate ops ((conv + bias) + shortcut, then
// max) -> identical bits; float16: float32 arithmetic, one rounding at the end.
#include <hip/hip_fp16.h>

#include "odet_internal.h"

struct EpiParams {
  void* x; const void* bias; const void* res;
  long long total;      // npix * (C / N) vectors
  int vec_per_px, relu;
};

template <typename FT>
__global__ void __launch_bounds__(256) k_bias_act(EpiParams p) {
  constexpr int N = sizeof(FT) == 2 ? 8 : 4;
  uint4* __restrict__ x = reinterpret_cast<uint4*>(p.x);
  const uint4* __restrict__ bias = reinterpret_cast<const uint4*>(p.bias);
  const uint4* __restrict__ res = reinterpret_cast<const uint4*>(p.res);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < p.total; i += (long long)gridDim.x * 256) {
    const uint4 xv = x[i];
    const uint4 bv = bias[(int)(i % p.vec_per_px)];
    uint4 rv = make_uint4(0, 0, 0, 0);
    if (res) rv = res[i];
    float a[8], b[8], r[8];
    if (sizeof(FT) == 2) {
      const uint32_t xw[4] = {xv.x, xv.y, xv.z, xv.w}, bw[4] = {bv.x, bv.y, bv.z, bv.w}, rw[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float2 t = __half22float2(*reinterpret_cast<const __half2*>(&xw[k]));
        const float2 u = __half22float2(*reinterpret_cast<const __half2*>(&bw[k]));
        const float2 v = __half22float2(*reinterpret_cast<const __half2*>(&rw[k]));
        a[2 * k] = t.x; a[2 * k + 1] = t.y; b[2 * k] = u.x; b[2 * k + 1] = u.y; r[2 * k] = v.x; r[2 * k + 1] = v.y;
      }
    } else {
      const uint32_t xw[4] = {xv.x, xv.y, xv.z, xv.w}, bw[4] = {bv.x, bv.y, bv.z, bv.w}, rw[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        a[k] = __uint_as_float(xw[k]); b[k] = __uint_as_float(bw[k]); r[k] = __uint_as_float(rw[k]);
      }
    }
#pragma unroll
    for (int k = 0; k < N; ++k) {
      float y = a[k] + b[k];
      if (res) y = y + r[k];
      if (p.relu) y = (y < 0.0f) ? 0.0f : y;   // (a NaN stays a NaN, as in the framework's relu)
      a[k] = y;
    }
    uint4 o;
    if (sizeof(FT) == 2) {
      uint32_t w[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const __half2 t = __floats2half2_rn(a[2 * k], a[2 * k + 1]);
        w[k] = *reinterpret_cast<const uint32_t*>(&t);
      }
      o = make_uint4(w[0], w[1], w[2], w[3]);
    } else {
      o = make_uint4(__float_as_uint(a[0]), __float_as_uint(a[1]), __float_as_uint(a[2]), __float_as_uint(a[3]));
    }
    x[i] = o;
  }
}

extern "C" int odet_bias_act(void* x, const void* bias, const void* residual, long long npix, int C, int relu,
                             int f16, odet_stream_t stream) {
  ODET_REQUIRE(x && bias, "odet_bias_act: null pointer");
  ODET_REQUIRE(npix >= 0 && C > 0, "odet_bias_act: bad sizes");
  const int n = f16 ? 8 : 4;
  ODET_REQUIRE(C % n == 0, "odet_bias_act: C must be a multiple of %d (got %d)", n, C);
  if (npix == 0) return ODET_OK;
  EpiParams p;
  p.x = x; p.bias = bias; p.res = residual;
  p.vec_per_px = C / n;
  p.total = npix * p.vec_per_px;
  p.relu = relu ? 1 : 0;
  const long long blocks = (p.total + 255) / 256;
  const int grid = (int)std::min<long long>(blocks, 256 * 64);
  if (f16)
    hipLaunchKernelGGL(k_bias_act<__half>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
  else
    hipLaunchKernelGGL(k_bias_act<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
  ODET_LAUNCH_CHECK();
  return ODET_OK;
}

// ---- convolution epilogue + max-pooling in one pass -------------------------------------------------------
// resnet_fpn.py:228-259 (conv1 -> bn -> relu -> pool1_pad + 3x3/2 max-pool) and vgg16_faster_rcnn.py:260-342
// (conv + relu, then MaxPooling2D((2,2), 2, 'same') after each stage): out = maxpool(relu(conv + bias)).
// Bias is per channel and ReLU / rounding are monotone, so maxpool(relu(x + b)) == relu(max(x) + b) bit for bit:
// the window maximum is taken over the RAW convolution output, one pass reads it once and writes the pooled map
// (1/4 of the pixels) -- instead of an epilogue pass (read + write of the full map) followed by a pooling pass.
// Window taps outside the map are skipped; with values >= 0 after the ReLU that equals the reference's zero
// padding (ZeroPadding2D + 'valid') and TF's 'same' pooling alike.
struct PoolEpiParams {
  const void* x; const void* bias; void* out;
  int B, H, W, C, OH, OW, k, stride, pad;
  long long total;            // B * OH * OW * (C / N) vectors
};

template <typename FT>
__global__ void __launch_bounds__(256) k_bias_relu_maxpool(PoolEpiParams p) {
  constexpr int N = sizeof(FT) == 2 ? 8 : 4;
  const uint4* __restrict__ x = reinterpret_cast<const uint4*>(p.x);
  const uint4* __restrict__ bias = reinterpret_cast<const uint4*>(p.bias);
  uint4* __restrict__ out = reinterpret_cast<uint4*>(p.out);
  const int vpp = p.C / N;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < p.total; i += (long long)gridDim.x * 256) {
    const int v = (int)(i % vpp);
    long long t = i / vpp;
    const int ox = (int)(t % p.OW); t /= p.OW;
    const int oy = (int)(t % p.OH);
    const int b = (int)(t / p.OH);
    float m[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) m[e] = -INFINITY;
    for (int ky = 0; ky < p.k; ++ky) {
      const int iy = oy * p.stride - p.pad + ky;
      if (iy < 0 || iy >= p.H) continue;
      for (int kx = 0; kx < p.k; ++kx) {
        const int ix = ox * p.stride - p.pad + kx;
        if (ix < 0 || ix >= p.W) continue;
        const uint4 xv = x[(((long long)b * p.H + iy) * p.W + ix) * vpp + v];
        const uint32_t xw[4] = {xv.x, xv.y, xv.z, xv.w};
        if (sizeof(FT) == 2) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float2 f = __half22float2(*reinterpret_cast<const __half2*>(&xw[q]));
            m[2 * q] = fmaxf(m[2 * q], f.x); m[2 * q + 1] = fmaxf(m[2 * q + 1], f.y);
          }
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            // (float32: a NaN stays -- the two-limb convolutions mark an operand beyond float16's range that way and the
            // detectors look for it at the end of the pass, model/fpn_detector.py range_ok; fmaxf would drop it)
            const float f = __uint_as_float(xw[q]);
            m[q] = (f > m[q] || f != f) ? f : m[q];
          }
        }
      }
    }
    const uint4 bv = bias[v];
    const uint32_t bw[4] = {bv.x, bv.y, bv.z, bv.w};
    uint4 o;
    if (sizeof(FT) == 2) {
      uint32_t w[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float2 f = __half22float2(*reinterpret_cast<const __half2*>(&bw[q]));
        float a0 = m[2 * q] + f.x, a1 = m[2 * q + 1] + f.y;
        a0 = (a0 < 0.0f) ? 0.0f : a0; a1 = (a1 < 0.0f) ? 0.0f : a1;
        const __half2 hh = __floats2half2_rn(a0, a1);
        w[q] = *reinterpret_cast<const uint32_t*>(&hh);
      }
      o = make_uint4(w[0], w[1], w[2], w[3]);
    } else {
      float a[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) { a[q] = m[q] + __uint_as_float(bw[q]); a[q] = (a[q] < 0.0f) ? 0.0f : a[q]; }
      o = make_uint4(__float_as_uint(a[0]), __float_as_uint(a[1]), __float_as_uint(a[2]), __float_as_uint(a[3]));
    }
    out[i] = o;
  }
}

extern "C" int odet_bias_relu_maxpool(const void* x, const void* bias, void* out, int B, int H, int W, int C, int OH,
                                      int OW, int kernel, int stride, int pad, int f16, odet_stream_t stream) {
  ODET_REQUIRE(x && bias && out, "odet_bias_relu_maxpool: null pointer");
  ODET_REQUIRE(B >= 0 && H > 0 && W > 0 && C > 0 && OH > 0 && OW > 0, "odet_bias_relu_maxpool: bad sizes");
  ODET_REQUIRE(kernel >= 1 && kernel <= 4 && stride >= 1 && pad >= 0 && pad < kernel, "odet_bias_relu_maxpool: bad window");
  // every output window must hold at least one pixel of the map
  ODET_REQUIRE((long long)(OH - 1) * stride - pad < H && (long long)(OW - 1) * stride - pad < W,
               "odet_bias_relu_maxpool: output larger than the pooled map");
  const int n = f16 ? 8 : 4;
  ODET_REQUIRE(C % n == 0, "odet_bias_relu_maxpool: C must be a multiple of %d (got %d)", n, C);
  if (B == 0) return ODET_OK;
  PoolEpiParams p;
  p.x = x; p.bias = bias; p.out = out;
  p.B = B; p.H = H; p.W = W; p.C = C; p.OH = OH; p.OW = OW; p.k = kernel; p.stride = stride; p.pad = pad;
  p.total = (long long)B * OH * OW * (C / n);
  const int grid = (int)std::min<long long>((p.total + 255) / 256, 256 * 64);
  if (f16)
    hipLaunchKernelGGL(k_bias_relu_maxpool<__half>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
  else
    hipLaunchKernelGGL(k_bias_relu_maxpool<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
  ODET_LAUNCH_CHECK();
  return ODET_OK;
}

// ---- RPN head epilogue ---------------------------------------------------------------
// base_fpn_model.py:188-200,427-432: the shared RpnHead runs on every pyramid level, its 1x1 convolutions emit
// [h,w,2A] scores and [h,w,4A] box deltas, reshaped to [-1,2] / [-1,4] and concatenated P2 -> P6.  In NHWC the
// reshape is the identity, so the "re-layout" is: add the bias, widen to float32, write at the level's offset of
// the concatenated arrays the proposal stage reads -- one pass per level instead of two bias adds, two casts and
// a share of two concatenations.
struct RpnPackParams {
  const void* in; const void* bias; float* out;
  long long per_image;        // h * w * ch values of one image of this level
  long long out_image_stride; // values of one image in the concatenated array (N * 2 or N * 4)
  long long out_offset;       // this level's first value inside an image
  long long total;            // B * per_image
  int ch;                     // 2A or 4A
};

template <typename FT>
__global__ void __launch_bounds__(256) k_rpn_pack(RpnPackParams p) {
  const FT* __restrict__ in = reinterpret_cast<const FT*>(p.in);
  const FT* __restrict__ bias = reinterpret_cast<const FT*>(p.bias);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < p.total; i += (long long)gridDim.x * 256) {
    const long long b = i / p.per_image, r = i - b * p.per_image;
    const float v = (float)in[i] + (float)bias[(int)(r % p.ch)];
    p.out[b * p.out_image_stride + p.out_offset + r] = v;
  }
}

extern "C" int odet_rpn_pack(const void* level_out, const void* bias, long long pixels, int ch, int B, float* out,
                             long long out_image_stride, long long out_offset, int f16, odet_stream_t stream) {
  ODET_REQUIRE(level_out && bias && out, "odet_rpn_pack: null pointer");
  ODET_REQUIRE(pixels >= 0 && ch > 0 && B >= 0 && out_image_stride >= 0 && out_offset >= 0, "odet_rpn_pack: bad sizes");
  ODET_REQUIRE(out_offset + pixels * ch <= out_image_stride, "odet_rpn_pack: level does not fit the concatenated array");
  RpnPackParams p;
  p.in = level_out; p.bias = bias; p.out = out;
  p.per_image = pixels * ch;
  p.out_image_stride = out_image_stride; p.out_offset = out_offset;
  p.total = (long long)B * p.per_image;
  p.ch = ch;
  if (p.total == 0) return ODET_OK;
  const int grid = (int)std::min<long long>((p.total + 255) / 256, 256 * 16);
  if (f16)
    hipLaunchKernelGGL(k_rpn_pack<__half>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
  else
    hipLaunchKernelGGL(k_rpn_pack<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
  ODET_LAUNCH_CHECK();
  return ODET_OK;
}

// The RpnHead's two 1x1 convolutions as ONE contraction (weights concatenated along the output channel: 2A
// score channels then 4A delta channels): the 512-channel activation is read once instead of twice, and this
// pass splits the [B, pixels, 6A] result into the two concatenated arrays.
struct RpnPackPairParams {
  const void* in; const void* bias; float* scores; float* deltas;
  long long per_image;            // pixels * 6A
  long long scores_image_stride, scores_offset, deltas_image_stride, deltas_offset;
  long long total;
  int A;
};

template <typename FT>
__global__ void __launch_bounds__(256) k_rpn_pack_pair(RpnPackPairParams p) {
  const FT* __restrict__ in = reinterpret_cast<const FT*>(p.in);
  const FT* __restrict__ bias = reinterpret_cast<const FT*>(p.bias);
  const int ch = 6 * p.A, sc = 2 * p.A;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < p.total; i += (long long)gridDim.x * 256) {
    const long long b = i / p.per_image, r = i - b * p.per_image;
    const long long px = r / ch;
    const int c = (int)(r - px * ch);
    const float v = (float)in[i] + (float)bias[c];
    if (c < sc)
      p.scores[b * p.scores_image_stride + p.scores_offset + px * sc + c] = v;
    else
      p.deltas[b * p.deltas_image_stride + p.deltas_offset + px * (ch - sc) + (c - sc)] = v;
  }
}

extern "C" int odet_rpn_pack_pair(const void* level_out, const void* bias, long long pixels, int A, int B, float* scores,
                                  long long scores_image_stride, long long scores_offset, float* deltas,
                                  long long deltas_image_stride, long long deltas_offset, int f16, odet_stream_t stream) {
  ODET_REQUIRE(level_out && bias && scores && deltas, "odet_rpn_pack_pair: null pointer");
  ODET_REQUIRE(pixels >= 0 && A > 0 && B >= 0 && scores_offset >= 0 && deltas_offset >= 0, "odet_rpn_pack_pair: bad sizes");
  ODET_REQUIRE(scores_offset + pixels * 2 * A <= scores_image_stride && deltas_offset + pixels * 4 * A <= deltas_image_stride,
               "odet_rpn_pack_pair: level does not fit the concatenated arrays");
  RpnPackPairParams p;
  p.in = level_out; p.bias = bias; p.scores = scores; p.deltas = deltas;
  p.per_image = pixels * 6 * A;
  p.scores_image_stride = scores_image_stride; p.scores_offset = scores_offset;
  p.deltas_image_stride = deltas_image_stride; p.deltas_offset = deltas_offset;
  p.total = (long long)B * p.per_image;
  p.A = A;
  if (p.total == 0) return ODET_OK;
  const int grid = (int)std::min<long long>((p.total + 255) / 256, 256 * 16);
  if (f16)
    hipLaunchKernelGGL(k_rpn_pack_pair<__half>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
  else
    hipLaunchKernelGGL(k_rpn_pack_pair<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
  ODET_LAUNCH_CHECK();
  return ODET_OK;
}

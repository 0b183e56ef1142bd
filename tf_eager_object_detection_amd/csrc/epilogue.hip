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

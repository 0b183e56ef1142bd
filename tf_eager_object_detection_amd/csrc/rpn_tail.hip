// RPN head tail on the matrix cores ("RPN head + score glue fused ... the re-layout disappears
// if the 1x1 epilogue writes ... in anchor order"): everything of the reference's RpnHead AFTER its 3x3 convolution
// (base_fpn_model.py:393-434, 188-200), for one pyramid level and a batch of images, in ONE pass over the
// 512-channel activation:
//     t      = relu(c + b1)                     c = the 3x3 convolution WITHOUT its bias, float16 NHWC [M, 512]
//     scores = t . Ws^T + bs   ([M, 2A])        -> float32 at the level's slice of the concatenated [B, N, 2] array
//     deltas = t . Wd^T + bd   ([M, 4A])        -> float32 at the level's slice of the concatenated [B, N, 4] array
// As separate ops this is an epilogue pass (read + write of [M, 512]), a 1x1 convolution (another read) and a pack
// pass; here c is read once and nothing else of that size moves: 2 * M * 512 bytes in, 24 * A * M bytes out.
//
// One wave owns 32 pixels and v_mfma_f32_32x32x16_f16 computes the transposed product (rows = output channels,
// columns = pixels) exactly as in conv1x1.hip: the pixel fragments are 16-byte global loads straight into operand
// registers (bias + ReLU applied on the way, b1 out of LDS), the 6A <= 24 weight rows sit in LDS as the rows of ONE
// 32-row MFMA block in an order that leaves lane-half 0 of a pixel with its 4A deltas and lane-half 1 with its 2A
// scores in consecutive registers -> contiguous float32 stores (48 + 24 bytes per pixel at A = 3).
#include <hip/hip_fp16.h>

#include "odet_internal.h"

typedef _Float16 rt_h8 __attribute__((ext_vector_type(8)));
typedef float rt_f16 __attribute__((ext_vector_type(16)));

struct RpnTailParams {
  const _Float16* c; const _Float16* b1; const _Float16* w; const _Float16* b2;   // w [6A, 512]: 2A score rows, 4A delta rows
  float* scores; float* deltas;
  long long M, px;                    // B * px pixels, px = pixels of the level in one image
  long long s_stride, s_off, d_stride, d_off;     // values per image / the level's first value in an image
  int A, slabs_per_wg;
};

constexpr int RT_K = 512, RT_KSTEPS = RT_K / 16, RT_LDW = RT_K + 8;

__global__ void __launch_bounds__(256, 2) k_rpn_tail_f16(RpnTailParams p) {
  __shared__ __align__(16) _Float16 wl[32 * RT_LDW];     // MFMA rows (permuted channels, zero rows for the unused)
  __shared__ __align__(16) _Float16 b1l[RT_K];
  __shared__ float b2l[32];                              // by MFMA row
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int A = p.A, nd = 4 * A, ns = 2 * A;
  // MFMA row i: lane-half (i >> 2) & 1, register 4 * (i >> 3) + (i & 3) -> half 0 = deltas, half 1 = scores
  for (int c = threadIdx.x; c < 32 * (RT_K / 8); c += 256) {
    const int row = c / (RT_K / 8), cc = c - row * (RT_K / 8);
    const int half = (row >> 2) & 1, reg = 4 * (row >> 3) + (row & 3);
    int src = -1;                                        // row of w
    if (half == 0 && reg < nd) src = ns + reg;
    if (half == 1 && reg < ns) src = reg;
    rt_h8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (_Float16)0.0f;
    if (src >= 0) v = *reinterpret_cast<const rt_h8*>(p.w + (long long)src * RT_K + 8 * cc);
    *reinterpret_cast<rt_h8*>(&wl[row * RT_LDW + 8 * cc]) = v;
  }
  if (threadIdx.x < RT_K / 8) *reinterpret_cast<rt_h8*>(&b1l[threadIdx.x * 8]) = *reinterpret_cast<const rt_h8*>(p.b1 + threadIdx.x * 8);
  if (threadIdx.x < 32) {
    const int row = threadIdx.x, half = (row >> 2) & 1, reg = 4 * (row >> 3) + (row & 3);
    float b = 0.0f;
    if (half == 0 && reg < nd) b = (float)p.b2[ns + reg];
    if (half == 1 && reg < ns) b = (float)p.b2[reg];
    b2l[row] = b;
  }
  __syncthreads();

  const _Float16* wrow = &wl[r * RT_LDW + 8 * h];
  const _Float16* brow = &b1l[8 * h];
  const long long slab0 = (long long)blockIdx.x * p.slabs_per_wg;
  for (int sidx = 0; sidx < p.slabs_per_wg; ++sidx) {
    const long long m = (slab0 + sidx) * 128 + wv * 32 + r;
    if ((slab0 + sidx) * 128 >= p.M) break;              // (uniform per workgroup; no barrier below)
    const long long mc = m < p.M ? m : p.M - 1;
    const _Float16* xrow = p.c + mc * RT_K + 8 * h;
    rt_f16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    // the pixel's fragments in four quarters of 8 k-steps: the next quarter is in flight while this one is used
    rt_h8 xq[2][8];
#pragma unroll
    for (int s = 0; s < 8; ++s) xq[0][s] = *reinterpret_cast<const rt_h8*>(xrow + 16 * s);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (q + 1 < 4) {
#pragma unroll
        for (int s = 0; s < 8; ++s) xq[(q + 1) & 1][s] = *reinterpret_cast<const rt_h8*>(xrow + 16 * (8 * (q + 1) + s));
      }
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const int ks = 8 * q + s;
        const rt_h8 bv = *reinterpret_cast<const rt_h8*>(brow + 16 * ks);
        const rt_h8 wv8 = *reinterpret_cast<const rt_h8*>(wrow + 16 * ks);
        rt_h8 t;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float v = (float)xq[q & 1][s][e] + (float)bv[e];     // bias_act's arithmetic: float32 add, ReLU, one rounding
          v = (v < 0.0f) ? 0.0f : v;
          t[e] = (_Float16)v;
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wv8, t, acc, 0, 0, 0);
      }
    }
    if (m < p.M) {
      const long long b = m / p.px, pi = m - b * p.px;
      // register reg of lane-half h = MFMA row (reg & 3) + 8 * (reg >> 2) + 4 * h
      if (h == 0) {
        float* o = p.deltas + b * p.d_stride + p.d_off + pi * nd;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg)
          if (reg < nd) o[reg] = acc[reg] + b2l[(reg & 3) + 8 * (reg >> 2)];
      } else {
        float* o = p.scores + b * p.s_stride + p.s_off + pi * ns;
#pragma unroll
        for (int reg = 0; reg < 8; ++reg)
          if (reg < ns) o[reg] = acc[reg] + b2l[(reg & 3) + 8 * (reg >> 2) + 4];
      }
    }
  }
}

extern "C" int odet_rpn_head_tail_f16(const void* conv_out, const void* conv_bias, const void* w, const void* bias,
                                      long long pixels, int A, int B, float* scores, long long scores_image_stride,
                                      long long scores_offset, float* deltas, long long deltas_image_stride,
                                      long long deltas_offset, odet_stream_t stream) {
  ODET_REQUIRE(conv_out && conv_bias && w && bias && scores && deltas, "odet_rpn_head_tail_f16: null pointer");
  ODET_REQUIRE(pixels >= 0 && B >= 0 && A >= 1 && A <= 4, "odet_rpn_head_tail_f16: bad sizes (1 <= A <= 4)");
  ODET_REQUIRE(scores_offset >= 0 && deltas_offset >= 0 && scores_offset + pixels * 2 * A <= scores_image_stride &&
               deltas_offset + pixels * 4 * A <= deltas_image_stride,
               "odet_rpn_head_tail_f16: level does not fit the concatenated arrays");
  ODET_REQUIRE(((uintptr_t)conv_out | (uintptr_t)conv_bias | (uintptr_t)w) % 16 == 0,
               "odet_rpn_head_tail_f16: pointers must be 16-byte aligned");
  const long long M = pixels * B;
  if (M == 0) return ODET_OK;
  RpnTailParams p;
  p.c = (const _Float16*)conv_out; p.b1 = (const _Float16*)conv_bias; p.w = (const _Float16*)w; p.b2 = (const _Float16*)bias;
  p.scores = scores; p.deltas = deltas;
  p.M = M; p.px = pixels;
  p.s_stride = scores_image_stride; p.s_off = scores_offset; p.d_stride = deltas_image_stride; p.d_off = deltas_offset;
  p.A = A;
  const long long slabs = (M + 127) / 128;
  // the 33 KB weight block is staged once per workgroup: a few slabs per workgroup on the big levels
  p.slabs_per_wg = slabs >= 4096 ? 4 : (slabs >= 1024 ? 2 : 1);
  const long long blocks = (slabs + p.slabs_per_wg - 1) / p.slabs_per_wg;
  ODET_REQUIRE(blocks < (1ll << 31), "odet_rpn_head_tail_f16: too many workgroups");
  hipLaunchKernelGGL(k_rpn_tail_f16, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
  ODET_LAUNCH_CHECK();
  return ODET_OK;
}

// Shared by the float32 forms of the implicit-GEMM convolution kernel: the exact-float32 matrix-instruction form
// (conv_f32.hip) and the split-precision form (conv_x3.hip: float32 operands as three bfloat16 limbs, six products per
// K-step on v_mfma_f32_16x16x32_bf16, float32 accumulation).  Both take ConvF32Params, leave a lane with pixel l15 of every
// 16-pixel tile and channels c0 + 16 t .. + 3 of tile t, and finish with the same epilogue.
#ifndef ODET_CONV_F32_COMMON_H_
#define ODET_CONV_F32_COMMON_H_
#include <hip/hip_runtime.h>

#include "odet_internal.h"

typedef float c3f4 __attribute__((ext_vector_type(4)));

struct ConvF32Params {
  const float* x[ODET_MAX_LEVELS]; float* y[ODET_MAX_LEVELS];
  const float* w; const float* bias;
  long long M[ODET_MAX_LEVELS];
  int H[ODET_MAX_LEVELS], W[ODET_MAX_LEVELS];
  long long tile_start[ODET_MAX_LEVELS + 1];
  int num_levels, cin, cout, relu;
  int tiles_n;
  // pointwise form (TAPS == 1, one map): output row m = (image, yo, xo) of a Ho x Wo map reads input pixel (yo, xo) * stride
  int stride, Ho, Wo;
  long long Min;
  const float* res;               // + shortcut [M][cout]
  const float* top; int th, tw; float tys, txs;   // or the FPN top-down merge: 0.5 * resize(top) + 0.5 * (conv + bias)
  const float* x2; int cin2, k1steps; long long Min2;   // or two sources along K ([x | x2(::stride)], weights concatenated)
  // split-K (conv_x3.hip only; 0 / 1 = off): ksplit consecutive workgroups share an output tile, each takes a contiguous part
  // of the K-steps and leaves its float32 partial tile in `part`; the one that draws the last of the tile's tickets adds the
  // parts in their fixed order 0 .. ksplit - 1 and runs the epilogue (deterministic; exact on integers)
  int ksplit; float* part; unsigned* ticket;
  float acc_scale;                // conv_x3.hip's two-limb float16 form: 2^-w_exp (the weight planes hold w * 2^w_exp); else unused
  // conv_x3.hip's two-limb form: the RANGE status word (nullable).  An activation beyond float16's range becomes an infinite
  // limb, and every product with it is infinite or NaN: every sum it enters is non-finite BEFORE bias / shortcut / ReLU, whatever
  // the weights' signs -- the epilogue ORs 1 into the word when it sees one (a wave ballot, then at most one atomic per wave)
  unsigned* status;
};

// bias (+ shortcut | FPN top-down merge) (+ ReLU) and the stores of a wave's MT x 4 accumulator tiles
template <int MT, int TAPS>
__device__ __forceinline__ void conv_f32_epilogue(const ConvF32Params& p, c3f4 (&acc)[MT][4], long long tile_m, int TM, int TN,
                                                  int wm, int wn, int tn, int l15, int lq, int lv, long long M, int cout) {
  // lane = pixel l15 of every pixel tile; tile t of the wave's 64-channel group: channels c0 + 16 t .. + 3
  const int c0 = tn * TN + wn * 64 + lq * 4;
  if (p.status) {
    // the raw sums, before anything can hide a non-finite one (ReLU maps -inf to 0): rows past M were computed from zeros
    bool bad = false;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) bad |= !(__builtin_fabsf(acc[mt][t][j]) <= 3.4028234663852886e38f);
    const unsigned long long any = __builtin_amdgcn_ballot_w64(bad);
    if (any != 0ull && (int)(threadIdx.x & 63) == (int)__builtin_ctzll(any)) atomicOr(p.status, 1u);
  }
  c3f4 bv[4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
    bv[t] = p.bias ? *reinterpret_cast<const c3f4*>(p.bias + c0 + 16 * t) : (c3f4){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const long long m = tile_m * TM + wm * 16 * MT + mt * 16 + l15;
    if (m < M) {
      float* dst = p.y[lv] + m * cout + c0;
      if constexpr (TAPS == 1) {
        if (p.top) {
          // the FPN top-down merge (neck.hip's arithmetic and operation order, float32 throughout: bit-identical to
          // odet_fpn_topdown_merge applied to the convolution's float32 result)
          const long long opx = (long long)p.Ho * p.Wo;
          const long long img = m / opx;
          const int rem = (int)(m - img * opx);
          const int yy = rem / p.Wo, xx = rem - yy * p.Wo;
          const float fy = (float)yy * p.tys, fx = (float)xx * p.txs;
          const float y0f = floorf(fy), x0f = floorf(fx);
          const int y0 = (int)y0f, x0 = (int)x0f;
          const int y1 = min(y0 + 1, p.th - 1), x1 = min(x0 + 1, p.tw - 1);
          const float yl = fy - y0f, xl = fx - x0f;
          const float* tb = p.top + (img * p.th * p.tw) * cout + c0;
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const c3f4 a = *reinterpret_cast<const c3f4*>(tb + ((long long)y0 * p.tw + x0) * cout + 16 * t);
            const c3f4 b = *reinterpret_cast<const c3f4*>(tb + ((long long)y0 * p.tw + x1) * cout + 16 * t);
            const c3f4 c = *reinterpret_cast<const c3f4*>(tb + ((long long)y1 * p.tw + x0) * cout + 16 * t);
            const c3f4 d = *reinterpret_cast<const c3f4*>(tb + ((long long)y1 * p.tw + x1) * cout + 16 * t);
            c3f4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float lat = acc[mt][t][j] + bv[t][j];
              const float tp = a[j] + (b[j] - a[j]) * xl;
              const float bt = c[j] + (d[j] - c[j]) * xl;
              const float up = tp + (bt - tp) * yl;
              o[j] = up * 0.5f + lat * 0.5f;
            }
            *reinterpret_cast<c3f4*>(dst + 16 * t) = o;
          }
          continue;
        }
      }
      const bool has_res = TAPS == 1 && p.res != nullptr;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        c3f4 o, r = (c3f4){0.0f, 0.0f, 0.0f, 0.0f};
        if (has_res) r = *reinterpret_cast<const c3f4*>(p.res + m * cout + c0 + 16 * t);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float v = acc[mt][t][j] + bv[t][j];
          if (has_res) v += r[j];
          if (p.relu) v = v < 0.0f ? 0.0f : v;
          o[j] = v;
        }
        *reinterpret_cast<c3f4*>(dst + 16 * t) = o;
      }
    }
  }
}

#endif  // ODET_CONV_F32_COMMON_H_

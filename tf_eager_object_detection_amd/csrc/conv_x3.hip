// Split-precision float32 forms of the implicit-GEMM convolution kernel: the detectors' float32 mode at the reference's
// accuracy on the bfloat16 matrix instructions (16 x the rate of the exact-float32 MFMA).  The reference computes its
// convolutions / dense layers in float32 (resnet_fpn.py:154-289, 339-407; base_fpn_model.py:393-434); here every float32
// operand is the EXACT sum of three bfloat16 limbs
//
//        a = a1 + a2 + a3,   a1 = bf16(a),  a2 = bf16(a - a1),  a3 = bf16(a - a1 - a2)        (round to nearest even; 3 x 8 = 24 bits)
//
// and a product is the six limb products down to 2^-16 of its size
//
//        a . b  ~  a1 b1 + (a1 b2 + a2 b1) + (a1 b3 + a2 b2 + a3 b1)         (dropped: a2 b3 + a3 b2 + a3 b3 <= 2^-23 |a b|, random sign)
//
// each exact in float32 (8 x 8 bits), summed in the MFMA's float32 accumulator: the result is within float32 rounding of
// the float64 truth, like the chain of float32 fused multiply-adds of v_mfma_f32_16x16x4_f32 (conv_f32.hip) -- a different,
// equally valid float32 evaluation of the same sum -- at 6 bfloat16 MFMAs per 32 k instead of 8 float32 MFMAs per 32 k of 1/16 the
// rate: 2.5 PFLOP/s / 6 = 417 "float32" TFLOP/s of peak against 157.
//
//  * WEIGHTS are split once on the host side (odet_split_bf16x3: three planes [3][cout][K] of bfloat16) and travel
//    global -> LDS by LDS-DMA, 64-byte rows.
//  * ACTIVATIONS stay float32 in memory (every layer reads and writes what conv_f32.hip does: a layer can run on either
//    form).  A K-step (one tap x 32 channels = 128 bytes per pixel) is loaded into REGISTERS (buffer loads, the zero padding
//    from the descriptor's range check), split there -- v_cvt_pk_bf16_f32 gives two limbs per instruction -- and written to
//    LDS as three planes while the matrix pipe works on the previous K-step.
//  * LDS image of a stage: [3 limbs][TM pixel rows][64 B] then [3 limbs][TN channel rows][64 B]; a fragment read takes 16
//    consecutive rows x 64 B.  ds_read_b128 is served in groups of 16 lanes ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ...:
//    MI355X guide, LDS table), and rows r and r + 4 of a 64-byte-row image share their banks: the 16-byte slot q of a row is
//    therefore stored at q ^ 2 for rows 8..15 of every 16 -- each group then covers all 64 banks once (without it every
//    fragment read is a 2-way conflict).
//  * Workgroup = 8 waves, tile TM = (8 / WN) * 16 * MT pixels x TN = 64 * WN channels, two stages (three limbs) / three (two limbs);
//    wave tile 16 MT pixels x 64 channels: per K-step NL x (4 + MT) fragment reads feed 24 MT / 12 MT MFMAs.
//  * Same transposed tiles, XCD-aware workgroup order, multi-level launches and epilogues (conv_f32_common.h) as the float32 kernel.
//  * The TWO-LIMB form (conv_tile_x3<.., NL = 2>, the odet_*_x2 entry points): float16 limbs h + l * 2^-11, three products per k;
//    see the template's header.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <type_traits>

#include "conv_f32_common.h"

typedef __bf16 x3b8 __attribute__((ext_vector_type(8)));
typedef __bf16 x3b2 __attribute__((ext_vector_type(2)));
typedef _Float16 x2h8 __attribute__((ext_vector_type(8)));
typedef _Float16 x2h2 __attribute__((ext_vector_type(2)));
typedef float x3f2 __attribute__((ext_vector_type(2)));
typedef unsigned int x3u4 __attribute__((ext_vector_type(4)));
typedef unsigned int x3u2 __attribute__((ext_vector_type(2)));
typedef __amdgpu_buffer_rsrc_t x3_rsrc_t;
typedef __attribute__((address_space(3))) void* x3_lds_ptr;

#define X3_BK 32                  // float32 input channels per K-step (128 bytes of a pixel's row; 64 bytes per limb)
#define X3_LDS_MAX (160 * 1024)
// K-steps the pixel slots are loaded ahead of their use, as a function of the LDS stages (register buffers; measured NS + 1 and
// NS + 2 on both forms: no gain -- what the slot path costs is its instructions, not its latency)
#ifndef X3_NA
#define X3_NA(NS) (NS)
#endif

// two float32 -> their three bfloat16 limbs, packed (low half = first value)
__device__ __forceinline__ void x3_split2(const x3f2 a, unsigned& h, unsigned& m, unsigned& l) {
  const x3b2 hb = __builtin_convertvector(a, x3b2);
  h = __builtin_bit_cast(unsigned, hb);
  const x3f2 hf = {__builtin_bit_cast(float, h << 16), __builtin_bit_cast(float, h & 0xFFFF0000u)};
  const x3f2 r1 = a - hf;                                 // exact (the difference has at most 16 significant bits)
  const x3b2 mb = __builtin_convertvector(r1, x3b2);
  m = __builtin_bit_cast(unsigned, mb);
  const x3f2 mf = {__builtin_bit_cast(float, m << 16), __builtin_bit_cast(float, m & 0xFFFF0000u)};
  const x3f2 r2 = r1 - mf;                                // exact (at most 8 significant bits are left)
  const x3b2 lb = __builtin_convertvector(r2, x3b2);
  l = __builtin_bit_cast(unsigned, lb);
}

// two float32 -> their two float16 limbs, packed: a ~ h + l * 2^-11, h = f16(a), l = f16((a - h) * 2^11) (the low limb scaled
// into float16's normal range; |a| > 65504 gives infinities -- and a NaN result -- never a wrong finite number)
__device__ __forceinline__ void x2_split2(const x3f2 a, unsigned& h, unsigned& l) {
  const x2h2 hb = __builtin_convertvector(a, x2h2);
  h = __builtin_bit_cast(unsigned, hb);
  const x3f2 hf = __builtin_convertvector(hb, x3f2);
  const x3f2 r = (a - hf) * 2048.0f;                      // exact (the difference has at most 13 significant bits)
  const x2h2 lb = __builtin_convertvector(r, x2h2);
  l = __builtin_bit_cast(unsigned, lb);
}

// ---- weights: float32 [n] -> planes [3][n] of bfloat16 (once per weight tensor) ------------------------------------------------
__global__ void __launch_bounds__(256) k_split_bf16x3(const float* __restrict__ w, unsigned* __restrict__ planes, long long n2) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n2; i += (long long)gridDim.x * 256) {
    const x3f2 a = {w[2 * i], w[2 * i + 1]};
    unsigned h, m, l;
    x3_split2(a, h, m, l);
    planes[i] = h; planes[n2 + i] = m; planes[2 * n2 + i] = l;
  }
}

extern "C" int odet_split_bf16x3(const float* w, void* planes, long long n, odet_stream_t stream) {
  ODET_REQUIRE(w && planes && n > 0 && n % 2 == 0, "odet_split_bf16x3: needs an even, positive element count");
  const long long n2 = n / 2;
  const int grid = (int)std::min<long long>((n2 + 255) / 256, 256 * 32);
  hipLaunchKernelGGL(k_split_bf16x3, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, (unsigned*)planes, n2);
  ODET_LAUNCH_CHECK();
  return ODET_OK;
}

// float32 [n] -> planes [2][n] of float16 of w * 2^w_exp (the caller picks w_exp so that the largest |w| lands in [512, 1024):
// every weight down to 2^-24 of the largest keeps both limbs in float16's normal range; the kernels scale the sums back)
__global__ void __launch_bounds__(256) k_split_f16x2(const float* __restrict__ w, unsigned* __restrict__ planes, long long n2, int w_exp) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n2; i += (long long)gridDim.x * 256) {
    const x3f2 a = {ldexpf(w[2 * i], w_exp), ldexpf(w[2 * i + 1], w_exp)};
    unsigned h, l;
    x2_split2(a, h, l);
    planes[i] = h; planes[n2 + i] = l;
  }
}

extern "C" int odet_split_f16x2(const float* w, void* planes, long long n, int w_exp, odet_stream_t stream) {
  ODET_REQUIRE(w && planes && n > 0 && n % 2 == 0, "odet_split_f16x2: needs an even, positive element count");
  ODET_REQUIRE(w_exp >= -100 && w_exp <= 100, "odet_split_f16x2: w_exp %d out of range", w_exp);
  const long long n2 = n / 2;
  const int grid = (int)std::min<long long>((n2 + 255) / 256, 256 * 32);
  hipLaunchKernelGGL(k_split_f16x2, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, (unsigned*)planes, n2, w_exp);
  ODET_LAUNCH_CHECK();
  return ODET_OK;
}

// ---- the tile ------------------------------------------------------------------------------------------------------------------
// p.w = the weight PLANES ([NL][cout][K] of 2-byte limbs, K = TAPS * cin (+ cin2)); everything else as in conv_f32.hip.
// NL = 3: bfloat16 limbs, six products (above).  NL = 2: float16 limbs h + l * 2^-11 (11 + 11 bits and the sign of l: 23 of
// float32's 24 bits -- every operand to within ONE float32 ulp, 2^-23), three products -- h h into one accumulator, h l + l h
// into a second one that joins it times 2^-11 at the end (dropped: l l <= 2^-22 |a b|): half the matrix work, for data inside
// float16's RANGE.  Against float64 on the network's layers its error is no larger than the exact-float32 form's
// (tools/r05/x3_layers.py --check: the accumulation's float32 rounding dominates both).
template <int MT, int WN, int TAPS, int NL>
__device__ __forceinline__ void conv_tile_x3(const ConvF32Params& p) {
  static_assert(NL == 2 || NL == 3, "two float16 or three bfloat16 limbs");
  using frag_t = std::conditional_t<NL == 3, x3b8, x2h8>;
  constexpr int WM = 8 / WN;
  constexpr int TM = WM * 16 * MT;
  constexpr int TN = 64 * WN;
  static_assert(TM % 64 == 0, "every wave loads whole 8-row pieces");
  constexpr int XP = TM / 64;                            // pixel pieces (8 rows x 128 B of float32) per wave and K-step
  constexpr int WPIECES = NL * TN / 16;                  // weight pieces (16 rows x 64 B) per stage
  constexpr int WPW = (WPIECES + 7) / 8;                 // per wave (the last ones only for the first waves)
  constexpr uint32_t XLIMB = (uint32_t)TM * 64u;         // bytes of one limb plane of the pixel rows
  constexpr uint32_t WLIMB = (uint32_t)TN * 64u;
  constexpr uint32_t WBASE = (uint32_t)NL * XLIMB;
  constexpr uint32_t STAGE = (uint32_t)NL * (XLIMB + WLIMB);
  // stages: the weights of K-step ks + NS - 1 travel while ks is computed.  Two limbs: three stages -- a K-step is 12 MFMAs per
  // pixel tile, shorter than an LDS-DMA's way through the memory system, so the copy gets two K-steps.
  constexpr int NS = NL == 2 ? 3 : 2;
  constexpr int NA = X3_NA(NS);                          // K-steps the pixel slots are loaded ahead (register buffers)
  static_assert(NS == 2 || WPIECES % 8 == 0, "the counted wait needs the same number of copies from every wave");
  extern __shared__ __align__(16) unsigned char lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wv / WN, wn = wv % WN;
  // split-K: `S` consecutive workgroups per output tile (host: S <= ksteps / 8, launches that would leave CUs idle)
  const int S = p.ksplit > 1 ? p.ksplit : 1;
  const long long blk = S > 1 ? (long long)(blockIdx.x / (unsigned)S) : (long long)blockIdx.x;
  const int z = S > 1 ? (int)(blockIdx.x - (unsigned)blk * (unsigned)S) : 0;
  const long long q8 = blk >> 3;
  const long long slab = (blk & 7) + 8 * (q8 / p.tiles_n);
  const int tn = (int)(q8 % p.tiles_n);
  if (slab >= p.tile_start[p.num_levels]) return;
  int lv = 0;
#pragma unroll
  for (int l = 1; l < ODET_MAX_LEVELS; ++l)
    if (l < p.num_levels && slab >= p.tile_start[l]) lv = l;
  const long long tile_m = slab - p.tile_start[lv];
  const int H = p.H[lv], W = p.W[lv], cin = p.cin, cout = p.cout;
  const uint32_t pixB = (uint32_t)cin * 4u;
  const uint32_t PAD = TAPS == 9 ? (uint32_t)(W + 1) * pixB : 0u;
  const uint32_t OOB = 0xFFFFFFF0u;
  const long long M = p.M[lv];
  const long long Min = TAPS == 9 ? M : p.Min;
  const x3_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(p.x[lv])) - PAD, 0, (int)((uint32_t)Min * pixB + 2u * PAD), 0x00020000);
  const bool dual = TAPS == 1 && p.x2 != nullptr;
  const uint32_t pixB2 = dual ? (uint32_t)p.cin2 * 4u : 0u;
  const x3_rsrc_t rx2 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(dual ? p.x2 : p.x[lv])), 0, (int)((uint32_t)(dual ? p.Min2 : 0) * pixB2), 0x00020000);
  const uint32_t Ktot = (uint32_t)TAPS * (uint32_t)cin + (dual ? (uint32_t)p.cin2 : 0u);
  const uint32_t wrowB = Ktot * 2u;                       // bytes of a weight row inside a limb plane
  const uint32_t planeB = (uint32_t)cout * wrowB;
  const x3_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, (int)((uint32_t)NL * planeB), 0x00020000);
  // ---- what this thread loads per K-step: XP 16-byte slots of float32 pixel rows (row = piece * 8 + lane / 8, slot lane % 8)
  const int sub = lane >> 3, sl = lane & 7;
  uint32_t voffA[XP], voffA2[TAPS == 1 ? XP : 1], maskA[XP];
#pragma unroll
  for (int i = 0; i < XP; ++i) {
    const int row = (wv + 8 * i) * 8 + sub;
    const long long m = tile_m * TM + row;
    uint32_t mk = 0;
    if constexpr (TAPS == 9) {
      if (m < M) {
        const long long img = m / ((long long)H * W);
        const int rem = (int)(m - img * H * W);
        const int yy = rem / W, xx = rem - yy * W;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int y2 = yy + t / 3 - 1, x2 = xx + t % 3 - 1;
          if (y2 >= 0 && y2 < H && x2 >= 0 && x2 < W) mk |= 1u << t;
        }
      }
      voffA[i] = (uint32_t)m * pixB + (uint32_t)sl * 16u;
    } else {
      long long src = m;
      if (p.stride != 1 && m < M) {
        const long long opx = (long long)p.Ho * p.Wo;
        const long long img = m / opx;
        const int rem = (int)(m - img * opx);
        const int yo = rem / p.Wo, xo = rem - yo * p.Wo;
        src = (img * H + (long long)yo * p.stride) * W + (long long)xo * p.stride;
      }
      mk = m < M ? 1u : 0u;
      voffA[i] = (uint32_t)(dual ? m : src) * pixB + (uint32_t)sl * 16u;
      voffA2[i] = (uint32_t)src * pixB2 + (uint32_t)sl * 16u;
    }
    maskA[i] = mk;
  }
  // its weight pieces: piece pi = wv + 8 i -> limb pi / (TN / 16), rows 16 (pi % (TN / 16)) + lane / 4, 16-byte slot lane % 4
  uint32_t voffW[WPW];
#pragma unroll
  for (int i = 0; i < WPW; ++i) {
    const int pi = wv + 8 * i;
    const int limb = pi / (TN / 16), rb = pi % (TN / 16);
    const int ch = tn * TN + rb * 16 + (lane >> 2);
    // (the DMA writes lane-linear: LDS slot lane % 4 of row lane / 4 receives the K slot (lane % 4) ^ 2 [rows 8..15])
    voffW[i] = pi < WPIECES ? (uint32_t)limb * planeB + (uint32_t)ch * wrowB + (uint32_t)((lane & 3) ^ ((lane >> 5) << 1)) * 16u : OOB;
  }
  const int chunks = cin / X3_BK;
  const int ksteps_all = TAPS * chunks + (dual ? p.cin2 / X3_BK : 0);
  const int k1steps = dual ? p.k1steps : ksteps_all;
  const int ks_lo = (int)((long long)ksteps_all * z / S);               // this workgroup's part of the K-steps
  const int ksteps = (int)((long long)ksteps_all * (z + 1) / S) - ks_lo;
  // the float32 slots of TWO K-steps in flight (buffer ks & 1): a load issued at the top of K-step ks is split and stored in
  // the middle of K-step ks + 1 -- one and a half K-steps (~3 us) to arrive, whatever level of the memory system it comes from
  x3u4 ra[NA][XP];
  auto load_a = [&](int ks, x3u4 (&r)[XP]) {
    int tap = 0;
    uint32_t soA;
    if constexpr (TAPS == 9) {
      tap = ks / chunks;
      const int chunk = ks - tap * chunks;
      soA = (uint32_t)((tap / 3) * W + tap % 3) * pixB + (uint32_t)chunk * 128u;
    } else {
      soA = (uint32_t)ks * 128u;
    }
    if (TAPS == 1 && ks >= k1steps) {
#pragma unroll
      for (int i = 0; i < XP; ++i)
        r[i] = __builtin_amdgcn_raw_buffer_load_b128(rx2, (int)((maskA[i] & 1u) ? voffA2[i] : OOB),
                                                     (int)((uint32_t)(ks - k1steps) * 128u), 0);
    } else {
#pragma unroll
      for (int i = 0; i < XP; ++i)
        r[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)(((maskA[i] >> tap) & 1u) ? voffA[i] : OOB), (int)soA, 0);
    }
  };
  auto issue_w = [&](int ks, uint32_t stage) {
#pragma unroll
    for (int i = 0; i < WPW; ++i)
      if (WPIECES % 8 == 0 || wv + 8 * i < WPIECES)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (x3_lds_ptr)(lds + stage + WBASE + (uint32_t)(wv + 8 * i) * 1024u), 16,
                                                 (int)voffW[i], (int)((uint32_t)ks * 64u), 0, 0);
  };
  // split a loaded slot and write its three limb planes: 4 channels = 8 bytes per limb
  // (row = piece * 8 + sub, piece = wv + 8 i: rows 8..15 of a 16-row tile are the odd pieces = the odd waves)
  const uint32_t sA = (uint32_t)sub * 64u + (uint32_t)(((sl >> 1) ^ ((wv & 1) << 1)) * 16 + (sl & 1) * 8);
  auto store_piece = [&](const x3u4 v, int i, uint32_t stage) {
    const c3f4 f = __builtin_bit_cast(c3f4, v);
    unsigned char* dst = lds + stage + (uint32_t)(wv + 8 * i) * 512u + sA;
    if constexpr (NL == 3) {
      unsigned h0, m0, l0, h1, m1, l1;
      x3_split2((x3f2){f[0], f[1]}, h0, m0, l0);
      x3_split2((x3f2){f[2], f[3]}, h1, m1, l1);
      const x3u2 h = {h0, h1}, m = {m0, m1}, l = {l0, l1};
      *reinterpret_cast<x3u2*>(dst) = h;
      *reinterpret_cast<x3u2*>(dst + XLIMB) = m;
      *reinterpret_cast<x3u2*>(dst + 2u * XLIMB) = l;
    } else {
      unsigned h0, l0, h1, l1;
      x2_split2((x3f2){f[0], f[1]}, h0, l0);
      x2_split2((x3f2){f[2], f[3]}, h1, l1);
      const x3u2 h = {h0, h1}, l = {l0, l1};
      *reinterpret_cast<x3u2*>(dst) = h;
      *reinterpret_cast<x3u2*>(dst + XLIMB) = l;
    }
  };
  const int l15 = lane & 15, lq = lane >> 4;
  const uint32_t fslot = (uint32_t)(lq ^ ((l15 >> 3) << 1)) * 16u;
  const uint32_t xoff = (uint32_t)(wm * 16 * MT + l15) * 64u + fslot;                          // + mt * 1024 + limb * XLIMB
  const uint32_t woff = WBASE + (uint32_t)(wn * 64 + l15) * 64u + fslot;                     // + t * 1024 + limb * WLIMB
  c3f4 acc[MT][4];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[mt][t] = (c3f4){0.0f, 0.0f, 0.0f, 0.0f};
  c3f4 acc2[NL == 2 ? MT : 1][4];                        // NL = 2: the h l + l h products (scaled by 2^11)
#pragma unroll
  for (int mt = 0; mt < (NL == 2 ? MT : 1); ++mt)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc2[mt][t] = (c3f4){0.0f, 0.0f, 0.0f, 0.0f};
  auto mma = [](const frag_t& a, const frag_t& b, const c3f4& c) {
    if constexpr (NL == 3) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  };
  constexpr int PPG = (XP + MT - 1) / MT;                // pixel pieces split and stored behind each group of MFMAs
  // the MFMAs of stage `sb`; behind pixel tile mt's 24 MFMAs the pieces mt * PPG .. of the NEXT K-step (registers `r`) are
  // split and stored into stage `nxt`: vector instructions and LDS stores issued in the shadow of the matrix pipe
  auto compute = [&](const unsigned char* sb, const x3u4 (&r)[XP], uint32_t nxt, bool store, auto&& early_issue, auto&& late_issue) {
    frag_t wf[NL][4];
#pragma unroll
    for (int l = 0; l < NL; ++l)
#pragma unroll
      for (int t = 0; t < 4; ++t) wf[l][t] = *reinterpret_cast<const frag_t*>(sb + woff + (uint32_t)l * WLIMB + (uint32_t)t * 1024u);
    frag_t xf[2][NL];
#pragma unroll
    for (int l = 0; l < NL; ++l) xf[0][l] = *reinterpret_cast<const frag_t*>(sb + xoff + (uint32_t)l * XLIMB);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int c = mt & 1;
      if (mt + 1 < MT) {
#pragma unroll
        for (int l = 0; l < NL; ++l)
          xf[c ^ 1][l] = *reinterpret_cast<const frag_t*>(sb + xoff + (uint32_t)l * XLIMB + (uint32_t)(mt + 1) * 1024u);
      }
      if (mt == 0) early_issue();                        // (behind the step's first fragment reads)
      // the products, smallest first; four independent accumulators between two MFMAs on the same one
      if constexpr (NL == 3) {
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[mt][t] = mma(wf[2][t], xf[c][0], acc[mt][t]);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[mt][t] = mma(wf[0][t], xf[c][2], acc[mt][t]);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[mt][t] = mma(wf[1][t], xf[c][1], acc[mt][t]);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[mt][t] = mma(wf[1][t], xf[c][0], acc[mt][t]);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[mt][t] = mma(wf[0][t], xf[c][1], acc[mt][t]);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[mt][t] = mma(wf[0][t], xf[c][0], acc[mt][t]);
      } else {
        const int m2 = NL == 2 ? mt : 0;
#pragma unroll
        for (int t = 0; t < 4; ++t) acc2[m2][t] = mma(wf[1][t], xf[c][0], acc2[m2][t]);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[mt][t] = mma(wf[0][t], xf[c][0], acc[mt][t]);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc2[m2][t] = mma(wf[0][t], xf[c][1], acc2[m2][t]);
      }
      if (mt == (MT > 1 ? 1 : 0)) late_issue();
      if (store) {
#pragma unroll
        for (int j = 0; j < PPG; ++j)
          if (mt * PPG + j < XP) store_piece(r[mt * PPG + j], mt * PPG + j, nxt);
      }
    }
  };
  // ---- K loop.  Step i: the weights of step i + NS - 1 travel into their stage (LDS-DMA), the pixel slots of step i + NS into
  // registers (buffer (i + NS) % NS = the one step i - 1 emptied); the slots of step i + 1 are split and stored between the MFMAs
  // of step i.  The counted wait before the barrier leaves everything younger than the weights of step i + 1 in flight; a bare
  // s_barrier, because __syncthreads() would drain it all.
  // The two waves of a SIMD (w and w + 4) run between the same barriers: waves 0-3 issue their copies right after the barrier,
  // waves 4-7 behind their second group of MFMAs, so that one of the two always has matrix work.  The two kinds run SEPARATE
  // copies of the loop (EARLY a compile-time constant), and the loop proper -- every step with all its copies to issue -- has
  // no condition in it: the compiler's wait-count insertion merges the pending loads of all paths it cannot tell apart, and
  // with both kinds in one body (or the end-of-K tests inside it) it drained EVERY outstanding copy (s_waitcnt vmcnt(0)) before
  // each refill of a slot buffer -- the prefetch depth was one step in name only (diagnostic build without the slot loads:
  // + 25 % three limbs, + 37 % two; tools/r05/x2_diag.sh).  The last 2 NS - 1 steps run the guarded form.
  // Slot buffers: NA = X3_NA(NS) register sets, the slots of step i + NA are loaded during step i (buffer i % NA, emptied by step
  // i - 1); the loop is unrolled NA steps (the buffers are registers: compile-time indices), the LDS stage of a step is a scalar.
  auto step = [&](auto early_c, auto steady_c, int i, uint32_t si, x3u4 (&rfree)[XP], const x3u4 (&rnext)[XP]) {
    constexpr bool EARLY = decltype(early_c)::value, STEADY = decltype(steady_c)::value;
    const int ks = ks_lo + i;
    const uint32_t cur = si * STAGE, nxt = (si + 1 == NS ? 0u : si + 1) * STAGE;
    const uint32_t wst = NS == 2 ? nxt : (si == 0 ? 2u : si - 1) * STAGE;          // stage of K-step i + NS - 1
    const bool more = STEADY || i + 1 < ksteps, morew = STEADY || i + NS - 1 < ksteps, morea = STEADY || i + NA < ksteps;
    // (the counted waits below count copies in ISSUE order: the slot loads are plain loads the compiler may move across the
    // LDS-DMA copies -- it did, once the loop was restructured --, so the order is pinned)
    auto issue = [&] {
      asm volatile("" ::: "memory");
      if (morew) issue_w(ks + NS - 1, wst);
      asm volatile("" ::: "memory");
      if (morea) load_a(ks + NA, rfree);
      asm volatile("" ::: "memory");
    };
    compute(lds + cur, rnext, nxt, more, [&] { if constexpr (EARLY) issue(); }, [&] { if constexpr (!EARLY) issue(); });
    // all but what was issued after the weights of step i + 1: the slots of that step (i - NS + 2) and, with three stages, the
    // weights and slots of step i
    if constexpr (NS == 2) {
      if (morea) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(XP) : "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    } else {
      const bool aprev = STEADY || i - 1 + NA < ksteps;
      if (morea) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(2 * XP + WPW) : "memory");
      else if (morew && aprev) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(XP + WPW) : "memory");
      else if (morew) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(WPW) : "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
  };
  auto kloop = [&](auto early_c) {
    constexpr std::true_type steady{};
    constexpr std::false_type guarded{};
    // ---- prologue, in the order the steps -NA .. -1 would have issued: step j the weights of j + NS - 1, then the slots of j + NA
#pragma unroll
    for (int j = -NA; j < 0; ++j) {
      asm volatile("" ::: "memory");
      if (j + NS - 1 >= 0 && j + NS - 1 < ksteps) issue_w(ks_lo + j + NS - 1, (uint32_t)(j + NS - 1) * STAGE);
      asm volatile("" ::: "memory");
      if (j + NA < ksteps) load_a(ks_lo + j + NA, ra[j + NA]);
    }
    asm volatile("" ::: "memory");
#pragma unroll
    for (int j = 0; j < XP; ++j) store_piece(ra[0][j], j, 0u);
    // (the weights of step 0 were issued by "step" 1 - NS: younger are its slots and everything of the later "steps")
    if constexpr (NS == 2) {
      if (ksteps > NA - 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(XP) : "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    } else {
      if (ksteps > NA - 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(2 * XP + WPW) : "memory");
      else if (ksteps > NA - 2 && ksteps > 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(XP + WPW) : "memory");
      else if (ksteps > 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(WPW) : "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    int i = 0;
    uint32_t si = 0;                                     // i % NS
    auto round = [&](auto steady_c) {                    // NA steps; buffers: step i + u refills u, splits and stores u + 1
#pragma unroll
      for (int u = 0; u < NA; ++u) {
        if (decltype(steady_c)::value || i + u < ksteps) {
          step(early_c, steady_c, i + u, si, ra[u], ra[u + 1 == NA ? 0 : u + 1]);
          si = si + 1 == NS ? 0u : si + 1;
        }
      }
    };
    for (; i + 2 * NA - 1 < ksteps; i += NA) round(steady);     // (the round's last step still has its slots of i + 2 NA - 1 to load)
    for (; i < ksteps; i += NA) round(guarded);
  };
  if (wv < 4) kloop(std::true_type{});
  else kloop(std::false_type{});
  if constexpr (NL == 2) {
    // the two accumulators joined, the weights' scale (a power of two) taken back out
    const float s2 = p.acc_scale * 0x1p-11f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[mt][t] = acc[mt][t] * p.acc_scale + acc2[mt][t] * s2;
  }
  if (S > 1) {
    // ---- split-K: leave this part's tile in the workspace, draw a ticket; the last of the tile's S workgroups adds the parts
    // in their fixed order and goes on to the epilogue.  (MI355X guide, inter-workgroup visibility: every storing wave drains
    // its stores, workgroup barrier, ONE lane releases at agent scope and only then signals; the consumer acquires at agent
    // scope before its loads -- the parts were written through other XCDs' L2s.)
    constexpr size_t PART = (size_t)8 * MT * 4 * 64;      // float4 values of one part: [wave][mt][t][lane]
    c3f4* base = reinterpret_cast<c3f4*>(p.part) + (size_t)blk * S * PART + (size_t)wv * (MT * 4 * 64) + lane;
    c3f4* mine = base + (size_t)z * PART;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int t = 0; t < 4; ++t) mine[(mt * 4 + t) * 64] = acc[mt][t];
    __shared__ int s_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      // atomicInc wraps to 0 at S - 1: the ticket is clean again for the next launch on this workspace
      s_last = atomicInc(p.ticket + blk, (unsigned)(S - 1)) == (unsigned)(S - 1) ? 1 : 0;
    }
    __syncthreads();
    if (!s_last) return;
    if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        c3f4 sum = __builtin_nontemporal_load(base + (mt * 4 + t) * 64);
        for (int zz = 1; zz < S; ++zz) sum += __builtin_nontemporal_load(base + (size_t)zz * PART + (mt * 4 + t) * 64);
        acc[mt][t] = sum;
      }
  }
  conv_f32_epilogue<MT, TAPS>(p, acc, tile_m, TM, TN, wm, wn, tn, l15, lq, lv, M, cout);
}

template <int MT, int WN>
__global__ void __launch_bounds__(512) k_conv3x3_x3(ConvF32Params p) {
  conv_tile_x3<MT, WN, 9, 3>(p);
}

template <int MT, int WN>
__global__ void __launch_bounds__(512) k_pointwise_x3(ConvF32Params p) {
  conv_tile_x3<MT, WN, 1, 3>(p);
}

template <int MT, int WN>
__global__ void __launch_bounds__(512) k_conv3x3_x2(ConvF32Params p) {
  conv_tile_x3<MT, WN, 9, 2>(p);
}

template <int MT, int WN>
__global__ void __launch_bounds__(512) k_pointwise_x2(ConvF32Params p) {
  conv_tile_x3<MT, WN, 1, 2>(p);
}

// ---- host side --------------------------------------------------------------------------------------------------------------
// tiles (MT, WN): 256 x 128, 128 x 128 | 128 x 256 | 256 x 64, 128 x 64, 64 x 64 (pixels x channels)
// (the two-limb form keeps two accumulator sets: 256 x 128 would not fit the register file)
#define X3_FOR_TILES(F) F(4, 2) F(2, 2) F(4, 4) F(2, 1) F(1, 1)
#define X2_FOR_TILES(F) F(2, 2) F(4, 4) F(2, 1) F(1, 1)

template <typename F>
static void x3_for_each_kernel(F f) {
#define X3_K(MT_, WN_) f((const void*)k_conv3x3_x3<MT_, WN_>); f((const void*)k_pointwise_x3<MT_, WN_>);
#define X2_K(MT_, WN_) f((const void*)k_conv3x3_x2<MT_, WN_>); f((const void*)k_pointwise_x2<MT_, WN_>);
  X3_FOR_TILES(X3_K)
  X2_FOR_TILES(X2_K)
#undef X3_K
#undef X2_K
}

static hipError_t x3_prepare_kernels() {
  static OdetPerDeviceOnce once;
  return once.run([] {
    hipError_t rc = hipSuccess;
    x3_for_each_kernel([&rc](const void* k) {
      // (the largest tile's two stages are 144 KB; the kernels also hold a few bytes of static LDS: the split-K flag)
      const hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
      if (e != hipSuccess) rc = e;
    });
    return rc;
  });
}

static unsigned x3_lds_bytes(int tm, int tn, int nl) { return (unsigned)((nl == 2 ? 3 : 2) * nl * (tm + tn) * 64); }
static int x3_tile_pixels(int mt, int wn) { return (8 / wn) * 16 * mt; }

#ifdef ODET_DIAG
// Diagnostic build only (-DODET_DIAG: tools/libodet_hip_diag.so, include/odet_diag.h; the shipped library has neither the entry
// point nor the override): force the tile and the K split of this process's split-precision launches -- (mt, wn) of
// X3_FOR_TILES, ksplit workgroups per tile (needs the workspace; not range-checked against the K-steps here); mt = 0 clears.
static std::atomic<unsigned> g_x3_override{0u};
extern "C" int odet_debug_x3_tile(int mt, int wn, int ksplit) {
  g_x3_override.store(mt > 0 ? ((unsigned)(ksplit > 1 ? ksplit : 1) << 16 | (unsigned)mt << 8 | (unsigned)wn) : 0u);
  return ODET_OK;
}
#endif

// ---- tile and split-K selection --------------------------------------------------------------------------------------------
// Cycles of a K-step on a CU, fitted to the tiles' measured rates on the ResNet-101-FPN layers (tools/r05/x3_tiles.py): the
// matrix work (6 MFMAs of 16 cycles per 16 x 16 tile on 4 SIMDs; two co-resident workgroups share the pipe) + what is not hidden
// behind it: ~3.5 cycles per pixel row (its float32 slots loaded, split and stored by the vector units), ~0.8 per weight row
// (LDS-DMA) and ~720 of barrier, waits and address arithmetic -- a pixel row costs four times a weight row, so the wide tile
// (128 pixels x 256 channels) wins wherever cout allows it.  A launch that would leave CUs idle (few pixels: batch 1 .. 8 on the
// 50 x 84 / 25 x 42 maps, the RoI head's dense layers) splits K over S workgroups per tile when every part keeps >= 8 K-steps:
// a deep-K, few-row layer is a latency chain of its K-steps (conv5's 3 x 3 at batch 1: 144 steps of 1.5 us on 18 workgroups).
#define X3_TICKETS 4096                                  // tiles of a split-K launch (the workspace's ticket words)
struct X3Pick { int mt, wn, ksplit; };
static X3Pick x3_pick_tile(const long long* M, int num_levels, int cout, int ksteps, size_t part_bytes_max, int nl) {
  static const int cand[][2] = {{4, 2}, {2, 2}, {4, 4}, {2, 1}, {1, 1}};       // (mt, wn)
  X3Pick best_pick{0, 0, 1};
  double best = 1e300;
  for (const auto& c : cand) {
    const int mt = c[0], wn = c[1];
    if (cout % (64 * wn) || (nl == 2 && mt == 4 && wn == 2)) continue;
    const int wm = 8 / wn, tm = wm * 16 * mt, tn = 64 * wn, tiles_n = cout / tn;
    long long slabs = 0;
    for (int l = 0; l < num_levels; ++l) slabs += (M[l] + tm - 1) / tm;
    const long long tiles = (slabs + 7) / 8 * 8 * tiles_n, real_tiles = slabs * tiles_n;
    const int occ = std::max(1, std::min(2, (int)(X3_LDS_MAX / x3_lds_bytes(tm, tn, nl))));
    // (two limbs: 3 products, two thirds of the weight rows' bytes, a shorter split)
    const double mfma = (double)tm * tn / 256.0 * (nl == 3 ? 6.0 : 3.0) * 16.0 / 4.0;
    const double other = (nl == 3 ? 3.5 : 3.0) * tm + (nl == 3 ? 0.8 : 0.55) * tn + 720.0;
    for (int S = 1; S <= 8; ++S) {
      // a split only where the tiles leave CUs idle (the parts of a launch that fills the chip would be HBM traffic of their
      // own: 1.1 GB for the RpnHead's P2 level at batch 1), every part at least 8 K-steps, tickets and parts inside the workspace
      if (S > 1 && (real_tiles > 384 || ksteps / S < 8 || tiles > X3_TICKETS || (size_t)tiles * S * tm * tn * 4 > part_bytes_max)) break;
      const long long blocks = real_tiles * S;
      const long long conc = std::min<long long>(occ, (blocks + 255) / 256);          // workgroups that share a CU
      const long long rounds = (blocks + 256 * conc - 1) / (256 * conc);
      // per workgroup: ~6000 cycles of prologue (first loads) and epilogue; with a split, the last workgroup of a tile reads the
      // S parts (~64 B / clock) behind a release / acquire hand-off (~2 us), and the parts are written and read once (~5 TB/s)
      const double reduce = S > 1 ? 4000.0 + (double)S * tm * tn / 16.0 : 0.0;
      const double traffic = S > 1 ? (double)real_tiles * S * tm * tn * 8.0 / 2400.0 : 0.0;
      const double cost = (double)rounds * ((double)((ksteps + S - 1) / S) * ((double)conc * mfma + other + 400.0) + 6000.0 + reduce) + traffic;
      if (cost < best * 0.97) { best = cost; best_pick = X3Pick{mt, wn, S}; }
    }
  }
#ifdef ODET_DIAG
  const unsigned o = g_x3_override.load();
  if (o && cout % (64 * (int)(o & 255)) == 0 && !(nl == 2 && (o >> 8 & 255) == 4 && (o & 255) == 2)) {
    best_pick = X3Pick{(int)(o >> 8 & 255), (int)(o & 255), std::max(1, (int)(o >> 16))};
    const int tm = x3_tile_pixels(best_pick.mt, best_pick.wn), tn = 64 * best_pick.wn;
    long long slabs = 0;
    for (int l = 0; l < num_levels; ++l) slabs += (M[l] + tm - 1) / tm;
    const long long tiles = (slabs + 7) / 8 * 8 * (cout / tn);
    if (best_pick.ksplit > ksteps || tiles > X3_TICKETS || (size_t)tiles * best_pick.ksplit * tm * tn * 4 > part_bytes_max)
      best_pick.ksplit = 1;                              // (a forced split that does not fit the workspace: none)
  }
#endif
  return best_pick;
}

// the workspace of the split-precision launches: X3_TICKETS ticket words (zero-filled ONCE by the caller; every launch leaves them
// zero), the two-limb form's RANGE status word (X3_STATUS_BYTES reserved; sticky: launches only ever OR into it, the caller
// reads and clears it), then the split-K parts
#define X3_STATUS_BYTES 64
#define X3_HEAD_BYTES ((size_t)X3_TICKETS * 4 + X3_STATUS_BYTES)
static int x3_apply_split(ConvF32Params* p, const X3Pick& pick, long long tiles, int TMsel, void* ws, size_t ws_bytes, const char* who) {
  p->ksplit = pick.ksplit;
  if (pick.ksplit <= 1) { p->ksplit = 0; return ODET_OK; }
  ODET_REQUIRE(ws && tiles <= X3_TICKETS && ws_bytes >= X3_HEAD_BYTES + (size_t)tiles * pick.ksplit * TMsel * 64 * pick.wn * 4,
               "%s: split-K workspace too small", who);
  p->ticket = (unsigned*)ws;
  p->part = (float*)((char*)ws + X3_HEAD_BYTES);
  return ODET_OK;
}
static size_t x3_part_bytes(const void* ws, size_t ws_bytes) {
  return (ws && ws_bytes > X3_HEAD_BYTES && (uintptr_t)ws % 16 == 0) ? ws_bytes - X3_HEAD_BYTES : 0;
}


template <bool PW>
static int x3_launch_tile(int nl, int wn, int mt, dim3 grid, unsigned lds_bytes, hipStream_t st, const ConvF32Params& p) {
#define X3_L(MT_, WN_)                                                                                      \
  if (nl == 3 && mt == MT_ && wn == WN_) {                                                                  \
    if (PW) hipLaunchKernelGGL((k_pointwise_x3<MT_, WN_>), grid, dim3(512), lds_bytes, st, p);              \
    else hipLaunchKernelGGL((k_conv3x3_x3<MT_, WN_>), grid, dim3(512), lds_bytes, st, p);                   \
    return ODET_OK;                                                                                         \
  }
#define X2_L(MT_, WN_)                                                                                      \
  if (nl == 2 && mt == MT_ && wn == WN_) {                                                                  \
    if (PW) hipLaunchKernelGGL((k_pointwise_x2<MT_, WN_>), grid, dim3(512), lds_bytes, st, p);              \
    else hipLaunchKernelGGL((k_conv3x3_x2<MT_, WN_>), grid, dim3(512), lds_bytes, st, p);                   \
    return ODET_OK;                                                                                         \
  }
  X3_FOR_TILES(X3_L)
  X2_FOR_TILES(X2_L)
#undef X3_L
#undef X2_L
  return odet_set_error(ODET_E_INVALID, "conv_x3: no kernel for the tile (mt %d, wn %d)", mt, wn);
}

static void x3_defaults(ConvF32Params* p) {
  p->stride = 1; p->Ho = p->Wo = 0; p->Min = 0; p->res = nullptr; p->top = nullptr; p->th = p->tw = 0; p->tys = p->txs = 0.0f;
  p->x2 = nullptr; p->cin2 = 0; p->k1steps = 0; p->Min2 = 0;
  p->ksplit = 0; p->part = nullptr; p->ticket = nullptr; p->acc_scale = 1.0f; p->status = nullptr;
}

// the limb form of a launch: 3 bfloat16 planes, or 2 float16 planes of w * 2^w_exp
struct X3Form { int nl, w_exp; };
static bool x3_form_ok(const X3Form& f) { return f.nl == 3 || (f.nl == 2 && f.w_exp >= -100 && f.w_exp <= 100); }
static unsigned* x3_status_word(const X3Form& form, void* ws, size_t ws_bytes) {
  return (form.nl == 2 && ws && ws_bytes >= X3_HEAD_BYTES && (uintptr_t)ws % 16 == 0) ? (unsigned*)((char*)ws + (size_t)X3_TICKETS * 4) : nullptr;
}

static int conv3x3_x3_launch(const X3Form& form, const odet_conv_level_t* levels, int num_levels, const void* w3, const void* bias,
                             int batch, int cin, int cout, int relu, void* ws, size_t ws_bytes, hipStream_t st) {
  ODET_REQUIRE(levels && w3, "odet_conv3x3_x3: null pointer");
  ODET_REQUIRE(x3_form_ok(form), "odet_conv3x3_x2: w_exp %d out of range", form.w_exp);
  ODET_REQUIRE(num_levels >= 1 && num_levels <= ODET_MAX_LEVELS, "odet_conv3x3_x3: num_levels %d out of range", num_levels);
  ODET_REQUIRE(batch > 0, "odet_conv3x3_x3: bad batch");
  ODET_REQUIRE(cin > 0 && cin % X3_BK == 0, "odet_conv3x3_x3: cin %d must be a multiple of %d", cin, X3_BK);
  ODET_REQUIRE(cout > 0 && cout % 64 == 0, "odet_conv3x3_x3: cout %d must be a multiple of 64", cout);
  ODET_REQUIRE((unsigned long long)cout * 9ull * cin * 6ull < 0x7FFFFFFFull, "odet_conv3x3_x3: weights too large");
  ODET_REQUIRE((uintptr_t)w3 % 16 == 0 && (uintptr_t)bias % 16 == 0, "odet_conv3x3_x3: pointers must be 16-byte aligned");
  ODET_HIP(x3_prepare_kernels());
  ConvF32Params p;
  x3_defaults(&p);
  for (int l = 0; l < ODET_MAX_LEVELS; ++l) {
    const odet_conv_level_t& L = levels[l < num_levels ? l : 0];
    ODET_REQUIRE(L.x && L.y && L.H > 0 && L.W > 0, "odet_conv3x3_x3: bad level %d", l);
    ODET_REQUIRE(((uintptr_t)L.x | (uintptr_t)L.y) % 16 == 0, "odet_conv3x3_x3: maps must be 16-byte aligned");
    const long long M = (long long)batch * L.H * L.W;
    ODET_REQUIRE((unsigned long long)M * cin * 4ull + 2ull * (L.W + 1) * cin * 4ull < 0xFFFFFFF0ull,
                 "odet_conv3x3_x3: level %d input larger than 4 GiB", l);
    p.x[l] = (const float*)L.x; p.y[l] = (float*)L.y; p.M[l] = M; p.H[l] = L.H; p.W[l] = L.W;
  }
  const X3Pick pick = x3_pick_tile(p.M, num_levels, cout, 9 * (cin / X3_BK), x3_part_bytes(ws, ws_bytes), form.nl);
  p.acc_scale = form.nl == 2 ? ldexpf(1.0f, -form.w_exp) : 1.0f;
  p.status = x3_status_word(form, ws, ws_bytes);
  const int wn = pick.wn, mt = pick.mt;
  const int TMsel = x3_tile_pixels(mt, wn);
  long long total = 0;
  for (int l = 0; l < ODET_MAX_LEVELS; ++l) {
    p.tile_start[l] = total;
    if (l < num_levels) total += (p.M[l] + TMsel - 1) / TMsel;
  }
  for (int l = num_levels; l <= ODET_MAX_LEVELS; ++l) p.tile_start[l] = total;
  p.w = (const float*)w3; p.bias = (const float*)bias;
  p.num_levels = num_levels; p.cin = cin; p.cout = cout; p.relu = relu ? 1 : 0;
  p.tiles_n = cout / (64 * wn);
  const long long blocks = (total + 7) / 8 * 8 * p.tiles_n;
  ODET_REQUIRE(blocks < (1ll << 28), "odet_conv3x3_x3: too many workgroups");
  const int rs = x3_apply_split(&p, pick, blocks, TMsel, ws, ws_bytes, "odet_conv3x3_x3");
  if (rs != ODET_OK) return rs;
  const int rc = x3_launch_tile<false>(form.nl, wn, mt, dim3((unsigned)(blocks * (p.ksplit > 1 ? p.ksplit : 1))),
                                       x3_lds_bytes(TMsel, 64 * wn, form.nl), st, p);
  if (rc != ODET_OK) return rc;
  ODET_LAUNCH_CHECK();
  return ODET_OK;
}

extern "C" size_t odet_x3_workspace_bytes(void) { return X3_HEAD_BYTES + ((size_t)64 << 20); }
extern "C" size_t odet_x2_status_offset(void) { return (size_t)X3_TICKETS * 4; }

extern "C" int odet_conv3x3_x3(const void* x, const void* w3, const void* bias, void* y, int batch, int H, int W, int cin,
                               int cout, int relu, void* workspace, size_t workspace_bytes, odet_stream_t stream) {
  ODET_REQUIRE(x && y, "odet_conv3x3_x3: null pointer");
  const odet_conv_level_t one{x, y, H, W};
  return conv3x3_x3_launch(X3Form{3, 0}, &one, 1, w3, bias, batch, cin, cout, relu, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int odet_conv3x3_x2(const void* x, const void* w2, const void* bias, void* y, int batch, int H, int W, int cin,
                               int cout, int relu, int w_exp, void* workspace, size_t workspace_bytes, odet_stream_t stream) {
  ODET_REQUIRE(x && y, "odet_conv3x3_x2: null pointer");
  const odet_conv_level_t one{x, y, H, W};
  return conv3x3_x3_launch(X3Form{2, w_exp}, &one, 1, w2, bias, batch, cin, cout, relu, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int odet_conv3x3_x2_levels(const odet_conv_level_t* levels, int num_levels, const void* w2, const void* bias,
                                      int batch, int cin, int cout, int relu, int w_exp, void* workspace, size_t workspace_bytes,
                                      odet_stream_t stream) {
  return conv3x3_x3_launch(X3Form{2, w_exp}, levels, num_levels, w2, bias, batch, cin, cout, relu, workspace, workspace_bytes,
                           (hipStream_t)stream);
}

extern "C" int odet_conv3x3_x3_levels(const odet_conv_level_t* levels, int num_levels, const void* w3, const void* bias,
                                      int batch, int cin, int cout, int relu, void* workspace, size_t workspace_bytes,
                                      odet_stream_t stream) {
  return conv3x3_x3_launch(X3Form{3, 0}, levels, num_levels, w3, bias, batch, cin, cout, relu, workspace, workspace_bytes,
                           (hipStream_t)stream);
}

struct PwX3Epilogue { const void* res; const void* top; int th, tw; const void* x2; int cin2; };

static int pointwise_x3_launch(const char* who, const X3Form& form, const void* x, const void* w3, const void* bias, void* y,
                               int batch, int H, int W, int stride, int cin, int cout, int relu, const PwX3Epilogue& epi, void* ws,
                               size_t ws_bytes, hipStream_t st) {
  ODET_REQUIRE(x && w3 && y, "%s: null pointer", who);
  ODET_REQUIRE(x3_form_ok(form), "%s: w_exp %d out of range", who, form.w_exp);
  ODET_REQUIRE(batch > 0 && H > 0 && W > 0 && (stride == 1 || stride == 2), "%s: bad shape", who);
  ODET_REQUIRE(cin % X3_BK == 0 && cin > 0, "%s: cin %d must be a positive multiple of %d", who, cin, X3_BK);
  ODET_REQUIRE(cout > 0 && cout % 64 == 0, "%s: cout %d must be a multiple of 64", who, cout);
  ODET_REQUIRE(((uintptr_t)x | (uintptr_t)w3 | (uintptr_t)y | (uintptr_t)bias | (uintptr_t)epi.res | (uintptr_t)epi.top |
                (uintptr_t)epi.x2) % 16 == 0, "%s: pointers must be 16-byte aligned", who);
  ODET_REQUIRE(!(epi.res && epi.top), "%s: shortcut and top-down merge exclude each other", who);
  ODET_REQUIRE(!epi.top || (stride == 1 && epi.th > 0 && epi.tw > 0 && !relu), "%s: bad merge arguments", who);
  ODET_HIP(x3_prepare_kernels());
  const int Ho = (H + stride - 1) / stride, Wo = (W + stride - 1) / stride;
  const long long M = (long long)batch * Ho * Wo;
  const long long Min = epi.x2 ? M : (long long)batch * H * W;
  ODET_REQUIRE((unsigned long long)Min * cin * 4ull < 0xFFFFFFF0ull, "%s: input larger than 4 GiB", who);
  ODET_REQUIRE(!epi.x2 || (epi.cin2 > 0 && epi.cin2 % X3_BK == 0 &&
                           (unsigned long long)batch * H * W * epi.cin2 * 4ull < 0xFFFFFFF0ull), "%s: bad second source", who);
  ODET_REQUIRE((unsigned long long)cout * (cin + (epi.x2 ? epi.cin2 : 0)) * 6ull < 0x7FFFFFFFull, "%s: weights too large", who);
  ConvF32Params p;
  x3_defaults(&p);
  for (int l = 0; l < ODET_MAX_LEVELS; ++l) {
    p.x[l] = (const float*)x; p.y[l] = (float*)y; p.M[l] = M; p.H[l] = H; p.W[l] = W;
  }
  p.res = (const float*)epi.res;
  p.top = (const float*)epi.top; p.th = epi.th; p.tw = epi.tw;
  p.tys = epi.top ? (float)epi.th / (float)Ho : 0.0f;
  p.txs = epi.top ? (float)epi.tw / (float)Wo : 0.0f;
  p.stride = stride; p.Ho = Ho; p.Wo = Wo; p.Min = Min;
  p.x2 = (const float*)epi.x2; p.cin2 = epi.x2 ? epi.cin2 : 0; p.k1steps = cin / X3_BK; p.Min2 = (long long)batch * H * W;
  const X3Pick pick = x3_pick_tile(&M, 1, cout, (cin + (epi.x2 ? epi.cin2 : 0)) / X3_BK, x3_part_bytes(ws, ws_bytes), form.nl);
  p.acc_scale = form.nl == 2 ? ldexpf(1.0f, -form.w_exp) : 1.0f;
  p.status = x3_status_word(form, ws, ws_bytes);
  const int wn = pick.wn, mt = pick.mt;
  const int TMsel = x3_tile_pixels(mt, wn);
  p.tiles_n = cout / (64 * wn);
  const long long total = (M + TMsel - 1) / TMsel;
  p.tile_start[0] = 0;
  for (int l = 1; l <= ODET_MAX_LEVELS; ++l) p.tile_start[l] = total;
  p.w = (const float*)w3; p.bias = (const float*)bias;
  p.num_levels = 1; p.cin = cin; p.cout = cout; p.relu = relu ? 1 : 0;
  const long long blocks = (total + 7) / 8 * 8 * p.tiles_n;
  ODET_REQUIRE(blocks < (1ll << 28), "%s: too many workgroups", who);
  const int rs = x3_apply_split(&p, pick, blocks, TMsel, ws, ws_bytes, who);
  if (rs != ODET_OK) return rs;
  const int rc = x3_launch_tile<true>(form.nl, wn, mt, dim3((unsigned)(blocks * (p.ksplit > 1 ? p.ksplit : 1))),
                                      x3_lds_bytes(TMsel, 64 * wn, form.nl), st, p);
  if (rc != ODET_OK) return rc;
  ODET_LAUNCH_CHECK();
  return ODET_OK;
}

extern "C" int odet_pointwise_x3(const void* x, const void* w3, const void* bias, const void* residual, void* y, int batch,
                                 int H, int W, int stride, int cin, int cout, int relu, void* workspace, size_t workspace_bytes,
                                 odet_stream_t stream) {
  const PwX3Epilogue e{residual, nullptr, 0, 0, nullptr, 0};
  return pointwise_x3_launch("odet_pointwise_x3", X3Form{3, 0}, x, w3, bias, y, batch, H, W, stride, cin, cout, relu, e, workspace,
                             workspace_bytes, (hipStream_t)stream);
}

extern "C" int odet_lateral_merge_x3(const void* x, const void* w3, const void* bias, const void* top, int th, int tw, void* y,
                                     int batch, int H, int W, int cin, int cout, void* workspace, size_t workspace_bytes,
                                     odet_stream_t stream) {
  ODET_REQUIRE(top, "odet_lateral_merge_x3: null pointer");
  const PwX3Epilogue e{nullptr, top, th, tw, nullptr, 0};
  return pointwise_x3_launch("odet_lateral_merge_x3", X3Form{3, 0}, x, w3, bias, y, batch, H, W, 1, cin, cout, 0, e, workspace,
                             workspace_bytes, (hipStream_t)stream);
}

extern "C" int odet_pointwise_dual_x3(const void* x1, int cin1, const void* x2, int cin2, int H2, int W2, int stride2,
                                      const void* w3, const void* bias, void* y, int batch, int cout, int relu,
                                      void* workspace, size_t workspace_bytes, odet_stream_t stream) {
  ODET_REQUIRE(x2, "odet_pointwise_dual_x3: null pointer");
  const PwX3Epilogue e{nullptr, nullptr, 0, 0, x2, cin2};
  return pointwise_x3_launch("odet_pointwise_dual_x3", X3Form{3, 0}, x1, w3, bias, y, batch, H2, W2, stride2, cin1, cout, relu, e,
                             workspace, workspace_bytes, (hipStream_t)stream);
}

// ---- the two-limb float16 forms: the same layers, `w2` = odet_split_f16x2's planes of w * 2^w_exp --------------------------
extern "C" int odet_pointwise_x2(const void* x, const void* w2, const void* bias, const void* residual, void* y, int batch,
                                 int H, int W, int stride, int cin, int cout, int relu, int w_exp, void* workspace,
                                 size_t workspace_bytes, odet_stream_t stream) {
  const PwX3Epilogue e{residual, nullptr, 0, 0, nullptr, 0};
  return pointwise_x3_launch("odet_pointwise_x2", X3Form{2, w_exp}, x, w2, bias, y, batch, H, W, stride, cin, cout, relu, e,
                             workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int odet_lateral_merge_x2(const void* x, const void* w2, const void* bias, const void* top, int th, int tw, void* y,
                                     int batch, int H, int W, int cin, int cout, int w_exp, void* workspace, size_t workspace_bytes,
                                     odet_stream_t stream) {
  ODET_REQUIRE(top, "odet_lateral_merge_x2: null pointer");
  const PwX3Epilogue e{nullptr, top, th, tw, nullptr, 0};
  return pointwise_x3_launch("odet_lateral_merge_x2", X3Form{2, w_exp}, x, w2, bias, y, batch, H, W, 1, cin, cout, 0, e, workspace,
                             workspace_bytes, (hipStream_t)stream);
}

extern "C" int odet_pointwise_dual_x2(const void* x1, int cin1, const void* x2, int cin2, int H2, int W2, int stride2,
                                      const void* w2, const void* bias, void* y, int batch, int cout, int relu, int w_exp,
                                      void* workspace, size_t workspace_bytes, odet_stream_t stream) {
  ODET_REQUIRE(x2, "odet_pointwise_dual_x2: null pointer");
  const PwX3Epilogue e{nullptr, nullptr, 0, 0, x2, cin2};
  return pointwise_x3_launch("odet_pointwise_dual_x2", X3Form{2, w_exp}, x1, w2, bias, y, batch, H2, W2, stride2, cin1, cout, relu,
                             e, workspace, workspace_bytes, (hipStream_t)stream);
}

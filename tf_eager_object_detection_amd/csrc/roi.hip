// RoI feature extraction: tf.image.crop_and_resize (bilinear, extrapolation 0) fused with the
// reference's 2x2 max / avg pooling, over all pyramid levels in one launch.
//
//   model/roi_pooling.py:45-90   RoiPoolingCropAndResize   (NORM_STRIDE, POOL_MAX2 | POOL_NONE)
//   model/roi_pooling.py:8-42    RoiPoolingCropAndResize2  (NORM_IMAGE,  POOL_MAX2)   <- FPN
//   model/roi_pooling.py:93-177  crop_and_resize/roi_align/RoiPoolingRoiAlign
//                                                          (NORM_TP_ALIGN, POOL_AVG2)
//
// The reference materialises the [R,14,14,C] crops (200 MB at R=1000, C=256) and pools them in
// a second op.  Here one workgroup produces one output ROW (P bins x C channels) of one RoI:
//
//   * the sampling arithmetic (box normalisation, per-sample coordinates) is evaluated once per
//     workgroup instead of once per bin;
//   * LDS-staged RoI tile: when the feature cells tapped by the row's S x (P*S) samples form a
//     small bounding tile (<= ROI_LDS_BYTES, i.e. sample spacing below ~1 cell -- the RoIs whose
//     bilinear taps overlap), the tile is loaded ONCE (NHWC: a cell's C channels are contiguous,
//     64 lanes x float4 = 1 KiB coalesced per cell) into LDS and all taps of the row are served
//     from there: 4-5x less L2 traffic than fetching every tap.  RoIs with wider spacing share no
//     taps; they read their taps straight from L2 (coalesced 1 KiB per tap).
//   * lerps in the exact TF operation order (no FMA), 2x2 max / avg reduced in registers -- max
//     is exact and the avg uses the same row-major sum, so results are bit-identical to the
//     un-fused form.
//
// Workgroup = 4 waves; wave w computes bins w, w+4, ... of the row, lane = 4 channels.
// Consecutive rows / (level-sorted) RoIs are mapped to the same XCD so that each XCD's L2 mostly
// holds one neighbourhood of one pyramid level.
#include <stdlib.h>

#include <algorithm>
#include <mutex>

#include <hip/hip_ext.h>
#include <hip/hip_fp16.h>

#include <type_traits>

#include "odet_internal.h"

#define ROI_LDS_BYTES (40 * 1024)   // staged tile budget: 4 workgroups per CU

struct RoiParams {
  const void* data[ODET_MAX_BATCH][ODET_MAX_LEVELS];    // [image of the batch][pyramid level]; float32 or float16
  PerImg<const float4*> rois;
  PerImg<const int32_t*> roi_level;
  PerImg<const int32_t*> count_dev;
  PerImg<const int32_t*> order;      // nullable: processing order of the RoIs (spatially sorted, odet_roi_order)
  PerImg<void*> out;
  int H[ODET_MAX_LEVELS];
  int W[ODET_MAX_LEVELS];
  float stride[ODET_MAX_LEVELS];
  int C, n, norm_mode, P, pool_mode, num_levels;
  float image_h, image_w;
  int nblocks;        // logical workgroups (before padding the grid to a multiple of 8)
  int blocks_per_xcd;
  int rows_per_wg;    // output rows of one RoI per workgroup
  int groups_per_roi; // ceil(P / rows_per_wg)
  int use_desc;       // whole-RoI descriptor form (roi_bins_desc)
  int xcd_images;     // 1: the image is derived from the XCD slot (batch of 2 / 4 / 8), 0: blockIdx.y
  int xcds_per_img;   // XCDs that serve one image (8 / batch)
  int f16;            // float16 feature maps / output
};

struct Axis {
  float start;   // in_(0)
  float scale;   // per-sample step
  float limit;   // dim - 1 (in sampled-map coordinates)
  float single;  // crop == 1: the one sample coordinate
};

// TF crop_and_resize_op.cc: in = lo_n * (dim-1) + i * scale, scale = (hi_n - lo_n)*(dim-1)/(crop-1)
__device__ __forceinline__ Axis make_axis(float lo_n, float hi_n, int dim, int crop) {
  Axis a;
  a.limit = (float)(dim - 1);
  a.scale = (crop > 1) ? (hi_n - lo_n) * a.limit / (float)(crop - 1) : 0.0f;
  a.start = lo_n * a.limit;
  a.single = 0.5f * (lo_n + hi_n) * a.limit;   // crop == 1 path
  return a;
}

__device__ __forceinline__ float axis_coord(const Axis& a, int i, int crop) {
  return (crop > 1) ? a.start + (float)i * a.scale : a.single;
}

struct Tap {      // one sample along one axis (wave-uniform)
  bool ok;        // TF: not extrapolated (0 <= in <= dim-1; NaN fails)
  int lo, hi;     // floor / ceil cell (after the SYMMETRIC-pad remap for the tensorpack modes)
  float lerp;
};

template <bool PAD>
__device__ __forceinline__ Tap make_tap(const Axis& a, int i, int crop, int dim) {
  Tap t;
  const float in = axis_coord(a, i, crop);
  // TF: extrapolate when (in < 0 || in > dim-1).  Written as the positive test so that a NaN
  // coordinate can never turn into a tap index.
  t.ok = (in >= 0.0f && in <= a.limit);
  const float f = floorf(in);
  t.lerp = in - f;
  int lo = (int)f, hi = (int)ceilf(in);
  if (PAD) {   // SYMMETRIC 1-px pad == edge replicate: padded[i] = src[clamp(i-1)]
    lo = min(max(lo - 1, 0), dim - 1);
    hi = min(max(hi - 1, 0), dim - 1);
  }
  t.lo = lo; t.hi = hi;
  return t;
}

__device__ __forceinline__ float4 lerp_tap(float4 tl, float4 tr, float4 bl, float4 br, float xw, float yw) {
  float4 r;
  float t, b;
  t = tl.x + (tr.x - tl.x) * xw; b = bl.x + (br.x - bl.x) * xw; r.x = t + (b - t) * yw;
  t = tl.y + (tr.y - tl.y) * xw; b = bl.y + (br.y - bl.y) * xw; r.y = t + (b - t) * yw;
  t = tl.z + (tr.z - tl.z) * xw; b = bl.z + (br.z - bl.z) * xw; r.z = t + (b - t) * yw;
  t = tl.w + (tr.w - tl.w) * xw; b = bl.w + (br.w - bl.w) * xw; r.w = t + (b - t) * yw;
  return r;
}

template <int POOL>
__device__ __forceinline__ float4 pool4(const float4 (&v)[2][2]) {
  float4 o;
  if (POOL == ODET_ROI_POOL_NONE) {
    o = v[0][0];
  } else if (POOL == ODET_ROI_POOL_MAX2) {
    o.x = fmaxf(fmaxf(v[0][0].x, v[0][1].x), fmaxf(v[1][0].x, v[1][1].x));
    o.y = fmaxf(fmaxf(v[0][0].y, v[0][1].y), fmaxf(v[1][0].y, v[1][1].y));
    o.z = fmaxf(fmaxf(v[0][0].z, v[0][1].z), fmaxf(v[1][0].z, v[1][1].z));
    o.w = fmaxf(fmaxf(v[0][0].w, v[0][1].w), fmaxf(v[1][0].w, v[1][1].w));
  } else {
    o.x = (((v[0][0].x + v[0][1].x) + v[1][0].x) + v[1][1].x) / 4.0f;
    o.y = (((v[0][0].y + v[0][1].y) + v[1][0].y) + v[1][1].y) / 4.0f;
    o.z = (((v[0][0].z + v[0][1].z) + v[1][0].z) + v[1][1].z) / 4.0f;
    o.w = (((v[0][0].w + v[0][1].w) + v[1][0].w) + v[1][1].w) / 4.0f;
  }
  return o;
}

// 4 consecutive channels of a cell: float32 maps as they are, float16 maps (BASELINE config 5) widened
// to float32 for the lerps and rounded to nearest-even on the way out
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ld4(const __half* p) {
  const uint2 u = *reinterpret_cast<const uint2*>(p);
  const __half2 a = *reinterpret_cast<const __half2*>(&u.x), b = *reinterpret_cast<const __half2*>(&u.y);
  const float2 fa = __half22float2(a), fb = __half22float2(b);
  return make_float4(fa.x, fa.y, fb.x, fb.y);
}
// Pooled features are written once and not read again by this path: non-temporal stores keep the 50 MB per
// image out of L2 / Infinity Cache, where they would evict the feature-map lines neighbouring RoIs share
// (measured in the bench: one-image launch 44-47 us -> 37-40 us with the maps streaming from HBM).
typedef float f4v __attribute__((ext_vector_type(4)));
typedef uint32_t u2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void st4(float* p, float4 v) {
  const f4v t = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(t, reinterpret_cast<f4v*>(p));
}
__device__ __forceinline__ void st4(__half* p, float4 v) {
  const __half2 a = __floats2half2_rn(v.x, v.y), b = __floats2half2_rn(v.z, v.w);
  u2v u;
  u.x = *reinterpret_cast<const uint32_t*>(&a);
  u.y = *reinterpret_cast<const uint32_t*>(&b);
  *reinterpret_cast<u2v*>(p) = u;      // (float16 outputs: plain stores; non-temporal measured neutral end to end)
}

__device__ __forceinline__ int rl_i(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ float rl_f(float v, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}

// One bin whose 2 x 2 samples are all inside the map, with its tapped cells DEDUPLICATED IN REGISTERS:
// along an axis the two samples tap cells (lo0, lo0+1) and (lo1, lo1+1); D = lo1 - lo0 in {0, 1} means
// they share cells, so only 2 + D distinct rows / columns are loaded (4, 6 or 9 cells instead of 16
// taps -- the RoIs whose sample spacing is below one cell, i.e. most of them).  D = 2 is the general
// form: (lo0, hi0, lo1, hi1) taken as they are, no sharing assumed.  Same values, same lerp
// arithmetic as the 16-tap form -> bit-identical results.
template <int POOL, int DY, int DX, typename FT>
__device__ __forceinline__ float4 roi_bin_shared(const FT* base, uint32_t C, uint32_t c, const uint32_t (&rowoff)[4],
                                                 const uint32_t (&col)[4], const float (&xw)[2],
                                                 const float (&yw)[2]) {
  constexpr int NR = (DY == 2) ? 4 : 2 + DY, NC = (DX == 2) ? 4 : 2 + DX;
  float4 blk[NR][NC];
#pragma unroll
  for (int i = 0; i < NR; ++i) {
#pragma unroll
    for (int j = 0; j < NC; ++j) {
#if defined(ODET_ROI_ABLATE) && (ODET_ROI_ABLATE == 1 || ODET_ROI_ABLATE == 5)   /* diagnostic: no loads (5: nor stores) */
      blk[i][j] = make_float4((float)(rowoff[i] + col[j]), xw[0], yw[0], (float)c);
#else
      blk[i][j] = ld4(base + (rowoff[i] + col[j]) * C + c);
#endif
    }
  }
#if defined(ODET_ROI_ABLATE) && ODET_ROI_ABLATE == 2     /* diagnostic: loads, no lerp arithmetic */
  {
    float4 acc = blk[0][0];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
#pragma unroll
      for (int j = 0; j < NC; ++j) acc.x = fmaxf(acc.x, blk[i][j].y + blk[i][j].z + blk[i][j].w + blk[i][j].x);
    }
    return acc;
  }
#endif
  float4 v[2][2];
#pragma unroll
  for (int sy = 0; sy < 2; ++sy) {
#pragma unroll
    for (int sx = 0; sx < 2; ++sx) {
      constexpr int dummy = 0;
      (void)dummy;
      const int rt = (DY == 2) ? 2 * sy : sy * DY, rb = rt + 1;
      const int cl = (DX == 2) ? 2 * sx : sx * DX, cr = cl + 1;
      v[sy][sx] = lerp_tap(blk[rt][cl], blk[rt][cr], blk[rb][cl], blk[rb][cr], xw[sx], yw[sy]);
    }
  }
  return pool4<POOL>(v);
}

// Bins w, w+4, ... of `nrows` consecutive output rows starting at row0 (w = wave).  The taps of ALL
// sample rows / columns were computed once, one per lane (tyl / txl: lane i = sample i); a bin fetches
// its S x S taps with v_readlane.  A tap (row y, col x) lives at float offset
// ((y - r0) * rs + (x - c0)) * C + c from `base` -- the feature map itself (r0 = c0 = 0, rs = W) or
// the LDS tile.
template <int POOL, bool PAD, bool LANE_TAPS>
__device__ __forceinline__ void roi_bins(const float* base, int rs, int r0, int c0, int C, int P, int crop,
                                         int Hdim, int Wdim, const Axis& ay, const Axis& ax, const Tap& tyl,
                                         const Tap& txl, int row0, int nrows, float* __restrict__ orow0) {
  constexpr int S = (POOL == ODET_ROI_POOL_NONE) ? 1 : 2;
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int txok = txl.ok ? 1 : 0, tyok = tyl.ok ? 1 : 0;
#if defined(ODET_ROI_ABLATE) && ODET_ROI_ABLATE == 4        /* diagnostic: prologue only */
  const int nbins = (tyl.lerp == 1.2345e30f) ? nrows * P : 0;
#else
  const int nbins = nrows * P;
#endif
  for (int b = w; b < nbins; b += 4) {
    const int pr = b / P;
    const int px = b - pr * P;
    const int py = row0 + pr;
    Tap tx[2], ty[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int xx = px * S + (s < S ? s : 0), yy = py * S + (s < S ? s : 0);
      if (LANE_TAPS) {
        tx[s].ok = rl_i(txok, xx) != 0; tx[s].lo = rl_i(txl.lo, xx); tx[s].hi = rl_i(txl.hi, xx);
        tx[s].lerp = rl_f(txl.lerp, xx);
        ty[s].ok = rl_i(tyok, yy) != 0; ty[s].lo = rl_i(tyl.lo, yy); ty[s].hi = rl_i(tyl.hi, yy);
        ty[s].lerp = rl_f(tyl.lerp, yy);
      } else {
        tx[s] = make_tap<PAD>(ax, xx, crop, Wdim);
        ty[s] = make_tap<PAD>(ay, yy, crop, Hdim);
        ty[s].lo = __builtin_amdgcn_readfirstlane(ty[s].lo);
        ty[s].hi = __builtin_amdgcn_readfirstlane(ty[s].hi);
      }
    }
    // row offsets (in cells) of the S sample rows, wave-uniform
    int rowlo[2], rowhi[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) { rowlo[s] = (ty[s].lo - r0) * rs; rowhi[s] = (ty[s].hi - r0) * rs; }
    float* __restrict__ obin = orow0 + ((size_t)pr * P + px) * C;
    if (S == 2 && !PAD && ty[0].ok && ty[1].ok && tx[0].ok && tx[1].ok) {
      // all four samples inside the map: deduplicated loads.  Register-sharing class per axis:
      // 0 / 1 = the two samples share cells, 2 = general
      int dy = 2, dx = 2;
      {
        const int d = ty[1].lo - ty[0].lo;
        if (ty[0].hi == ty[0].lo + 1 && ty[1].hi == ty[1].lo + 1 && (d == 0 || d == 1)) dy = d;
        const int e = tx[1].lo - tx[0].lo;
        if (tx[0].hi == tx[0].lo + 1 && tx[1].hi == tx[1].lo + 1 && (e == 0 || e == 1)) dx = e;
      }
      uint32_t rowoff[4], col[4];
      if (dy == 2) {
        rowoff[0] = (uint32_t)rowlo[0]; rowoff[1] = (uint32_t)rowhi[0];
        rowoff[2] = (uint32_t)rowlo[1]; rowoff[3] = (uint32_t)rowhi[1];
      } else {
        rowoff[0] = (uint32_t)rowlo[0]; rowoff[1] = rowoff[0] + (uint32_t)rs;
        rowoff[2] = rowoff[1] + (uint32_t)rs; rowoff[3] = rowoff[2];
      }
      if (dx == 2) {
        col[0] = (uint32_t)(tx[0].lo - c0); col[1] = (uint32_t)(tx[0].hi - c0);
        col[2] = (uint32_t)(tx[1].lo - c0); col[3] = (uint32_t)(tx[1].hi - c0);
      } else {
        col[0] = (uint32_t)(tx[0].lo - c0); col[1] = col[0] + 1; col[2] = col[0] + 2; col[3] = col[2];
      }
      const float xw[2] = {tx[0].lerp, tx[1].lerp};
      const float yw[2] = {ty[0].lerp, ty[1].lerp};
      const int cls = dy * 3 + dx;
      for (int c = lane * 4; c < C; c += 256) {
        float4 o;
        switch (cls) {
          case 0: o = roi_bin_shared<POOL, 0, 0>(base, (uint32_t)C, (uint32_t)c, rowoff, col, xw, yw); break;
          case 1: o = roi_bin_shared<POOL, 0, 1>(base, (uint32_t)C, (uint32_t)c, rowoff, col, xw, yw); break;
          case 2: o = roi_bin_shared<POOL, 0, 2>(base, (uint32_t)C, (uint32_t)c, rowoff, col, xw, yw); break;
          case 3: o = roi_bin_shared<POOL, 1, 0>(base, (uint32_t)C, (uint32_t)c, rowoff, col, xw, yw); break;
          case 4: o = roi_bin_shared<POOL, 1, 1>(base, (uint32_t)C, (uint32_t)c, rowoff, col, xw, yw); break;
          case 5: o = roi_bin_shared<POOL, 1, 2>(base, (uint32_t)C, (uint32_t)c, rowoff, col, xw, yw); break;
          case 6: o = roi_bin_shared<POOL, 2, 0>(base, (uint32_t)C, (uint32_t)c, rowoff, col, xw, yw); break;
          case 7: o = roi_bin_shared<POOL, 2, 1>(base, (uint32_t)C, (uint32_t)c, rowoff, col, xw, yw); break;
          default: o = roi_bin_shared<POOL, 2, 2>(base, (uint32_t)C, (uint32_t)c, rowoff, col, xw, yw); break;
        }
#if defined(ODET_ROI_ABLATE) && ODET_ROI_ABLATE == 3      /* diagnostic: no stores */
        if (o.x == 1.2345e30f) st4(obin + c, o);
#else
        st4(obin + c, o);
#endif
      }
      continue;
    }
    // general form: every sample guarded (extrapolated samples are 0), 4 taps each
    for (int c = lane * 4; c < C; c += 256) {
      float4 v[2][2];
#pragma unroll
      for (int sy = 0; sy < S; ++sy) {
#pragma unroll
        for (int sx = 0; sx < S; ++sx) {
          float4 res = make_float4(0, 0, 0, 0);
          if (ty[sy].ok && tx[sx].ok) {
            const uint32_t xl = (uint32_t)(tx[sx].lo - c0), xr = (uint32_t)(tx[sx].hi - c0);
            const uint32_t otl = ((uint32_t)rowlo[sy] + xl) * (uint32_t)C + (uint32_t)c;
            const uint32_t otr = ((uint32_t)rowlo[sy] + xr) * (uint32_t)C + (uint32_t)c;
            const uint32_t obl = ((uint32_t)rowhi[sy] + xl) * (uint32_t)C + (uint32_t)c;
            const uint32_t obr = ((uint32_t)rowhi[sy] + xr) * (uint32_t)C + (uint32_t)c;
            const float4 tl = *reinterpret_cast<const float4*>(base + otl);
            const float4 tr = *reinterpret_cast<const float4*>(base + otr);
            const float4 bl = *reinterpret_cast<const float4*>(base + obl);
            const float4 br = *reinterpret_cast<const float4*>(base + obr);
            res = lerp_tap(tl, tr, bl, br, tx[sx].lerp, ty[sy].lerp);
          }
          v[sy][sx] = res;
        }
      }
      st4(obin + c, pool4<POOL>(v));
    }
  }
}

// Whole-RoI form for the pooled modes with P*P <= 64 bins (un-padded): every lane first builds the
// DESCRIPTOR of one bin -- its four taps, the register-sharing class, the cell offsets of the rows /
// columns it loads and the lerp weights -- so the per-bin scalar bookkeeping is done once, 64 bins in
// parallel on the vector unit; a wave then walks its bins (wave, wave + nwaves, ...) and only fetches a
// descriptor with v_readlane before loading / lerping / storing.
template <int POOL, typename FT>
__device__ __forceinline__ void roi_bins_desc(const FT* base, int W, int C, int P, int crop, const Axis& ay,
                                              const Axis& ax, FT* __restrict__ oroi) {
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nwaves = blockDim.x >> 6;
  const int nbins = P * P;
  // ---- descriptor of bin `lane`
  const int bpy = lane / P, bpx = lane - bpy * P;
  const Tap ty0 = make_tap<false>(ay, min(2 * bpy, crop - 1), crop, 0);
  const Tap ty1 = make_tap<false>(ay, min(2 * bpy + 1, crop - 1), crop, 0);
  const Tap tx0 = make_tap<false>(ax, min(2 * bpx, crop - 1), crop, 0);
  const Tap tx1 = make_tap<false>(ax, min(2 * bpx + 1, crop - 1), crop, 0);
  const int okbits = (ty0.ok ? 1 : 0) | (ty1.ok ? 2 : 0) | (tx0.ok ? 4 : 0) | (tx1.ok ? 8 : 0);
  int dy = 2, dx = 2;
  {
    const int d = ty1.lo - ty0.lo;
    if (ty0.hi == ty0.lo + 1 && ty1.hi == ty1.lo + 1 && (d == 0 || d == 1)) dy = d;
    const int e = tx1.lo - tx0.lo;
    if (tx0.hi == tx0.lo + 1 && tx1.hi == tx1.lo + 1 && (e == 0 || e == 1)) dx = e;
  }
  const int cls_l = (okbits == 15) ? dy * 3 + dx : 9;          // 9 = general guarded form
  int ro0 = ty0.lo * W, ro1 = ty0.hi * W, ro2 = ty1.lo * W, ro3 = ty1.hi * W;
  int co0 = tx0.lo, co1 = tx0.hi, co2 = tx1.lo, co3 = tx1.hi;
  if (cls_l != 9) {
    if (dy != 2) { ro1 = ro0 + W; ro2 = ro1 + W; ro3 = ro2; }
    if (dx != 2) { co1 = co0 + 1; co2 = co0 + 2; co3 = co2; }
  }
  const float xw0 = tx0.lerp, xw1 = tx1.lerp, yw0 = ty0.lerp, yw1 = ty1.lerp;

#if defined(ODET_ROI_ABLATE) && ODET_ROI_ABLATE == 4        /* diagnostic: prologue + descriptors only */
  const int nb_run = (xw0 == 1.2345e30f) ? nbins : 0;
#else
  const int nb_run = nbins;
#endif
  for (int b = w; b < nb_run; b += nwaves) {
    const int cls = rl_i(cls_l, b);
    const uint32_t rowoff[4] = {(uint32_t)rl_i(ro0, b), (uint32_t)rl_i(ro1, b), (uint32_t)rl_i(ro2, b),
                                (uint32_t)rl_i(ro3, b)};
    const uint32_t col[4] = {(uint32_t)rl_i(co0, b), (uint32_t)rl_i(co1, b), (uint32_t)rl_i(co2, b),
                             (uint32_t)rl_i(co3, b)};
    const float xw[2] = {rl_f(xw0, b), rl_f(xw1, b)};
    const float yw[2] = {rl_f(yw0, b), rl_f(yw1, b)};
    FT* __restrict__ obin = oroi + (size_t)b * C;
    if (cls != 9) {
      for (int c = lane * 4; c < C; c += 256) {
        float4 o;
        switch (cls) {
          case 0: o = roi_bin_shared<POOL, 0, 0>(base, (uint32_t)C, (uint32_t)c, rowoff, col, xw, yw); break;
          case 1: o = roi_bin_shared<POOL, 0, 1>(base, (uint32_t)C, (uint32_t)c, rowoff, col, xw, yw); break;
          case 2: o = roi_bin_shared<POOL, 0, 2>(base, (uint32_t)C, (uint32_t)c, rowoff, col, xw, yw); break;
          case 3: o = roi_bin_shared<POOL, 1, 0>(base, (uint32_t)C, (uint32_t)c, rowoff, col, xw, yw); break;
          case 4: o = roi_bin_shared<POOL, 1, 1>(base, (uint32_t)C, (uint32_t)c, rowoff, col, xw, yw); break;
          case 5: o = roi_bin_shared<POOL, 1, 2>(base, (uint32_t)C, (uint32_t)c, rowoff, col, xw, yw); break;
          case 6: o = roi_bin_shared<POOL, 2, 0>(base, (uint32_t)C, (uint32_t)c, rowoff, col, xw, yw); break;
          case 7: o = roi_bin_shared<POOL, 2, 1>(base, (uint32_t)C, (uint32_t)c, rowoff, col, xw, yw); break;
          default: o = roi_bin_shared<POOL, 2, 2>(base, (uint32_t)C, (uint32_t)c, rowoff, col, xw, yw); break;
        }
#if defined(ODET_ROI_ABLATE) && (ODET_ROI_ABLATE == 3 || ODET_ROI_ABLATE == 5)     /* diagnostic: no stores */
        if (o.x == 1.2345e30f) st4(obin + c, o);
#else
        st4(obin + c, o);
#endif
      }
    } else {
      // general form: every sample guarded (extrapolated samples are 0), 4 taps each
      const int ok = rl_i(okbits, b);
      for (int c = lane * 4; c < C; c += 256) {
        float4 v[2][2];
#pragma unroll
        for (int sy = 0; sy < 2; ++sy) {
#pragma unroll
          for (int sx = 0; sx < 2; ++sx) {
            float4 res = make_float4(0, 0, 0, 0);
            if (((ok >> sy) & 1) && ((ok >> (2 + sx)) & 1)) {
              const uint32_t otl = (rowoff[2 * sy] + col[2 * sx]) * (uint32_t)C + (uint32_t)c;
              const uint32_t otr = (rowoff[2 * sy] + col[2 * sx + 1]) * (uint32_t)C + (uint32_t)c;
              const uint32_t obl = (rowoff[2 * sy + 1] + col[2 * sx]) * (uint32_t)C + (uint32_t)c;
              const uint32_t obr = (rowoff[2 * sy + 1] + col[2 * sx + 1]) * (uint32_t)C + (uint32_t)c;
              const float4 tl = ld4(base + otl);
              const float4 tr = ld4(base + otr);
              const float4 bl = ld4(base + obl);
              const float4 br = ld4(base + obr);
              res = lerp_tap(tl, tr, bl, br, xw[sx], yw[sy]);
            }
            v[sy][sx] = res;
          }
        }
        st4(obin + c, pool4<POOL>(v));
      }
    }
  }
}

// NORM: ODET_ROI_NORM_*; STAGE: allow the LDS-staged tile path.
template <int POOL, int NORM, bool STAGE, typename FT>
__global__ void __launch_bounds__(512) k_roi_pool(RoiParams p) {
  extern __shared__ __align__(16) float tile[];
  constexpr bool PAD = (NORM == ODET_ROI_NORM_TP_ALIGN);
  // XCD-aware remap: hardware deals workgroups round-robin over the 8 XCDs (each has its own L2).
  // A batch of 1 / 2 / 4 / 8 images gives every image 8 / 4 / 2 / 1 XCDs of its own, so that an XCD's L2
  // only ever holds lines of one image's maps; inside an image consecutive (level-sorted) RoIs share an XCD.
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int img = p.xcd_images ? xcd / p.xcds_per_img : blockIdx.y;
  const int sub = p.xcd_images ? xcd - img * p.xcds_per_img : xcd;
  const int lb = sub * p.blocks_per_xcd + slot;
  if (slot >= p.blocks_per_xcd || lb >= p.nblocks) return;
  const float4* __restrict__ rois = p.rois.v[img];
  const int32_t* __restrict__ roi_level = p.roi_level.v[img];
  const int32_t* __restrict__ count_dev = p.count_dev.v[img];
  FT* __restrict__ out = reinterpret_cast<FT*>(p.out.v[img]);
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int P = p.P, C = p.C;
  const int gpr = p.groups_per_roi;                 // workgroups per RoI (row groups of p.rows_per_wg rows)
  const int32_t* __restrict__ order = p.order.v[img];
  const int ri = lb / gpr;
  const int r = order ? min(max(order[ri], 0), p.n - 1) : ri;
  const int row0 = (lb - ri * gpr) * p.rows_per_wg;
  const int nrows = min(p.rows_per_wg, P - row0);
  FT* __restrict__ orow = out + ((size_t)r * P + row0) * P * C;

  // the three loads of the prologue are independent (r < n always addresses valid rows): one memory
  // latency instead of a chain of three
  const int cnt_raw = count_dev ? *count_dev : p.n;
  const int lvl_raw = roi_level ? roi_level[r] : 0;
  const float4 roi = rois[r];
  const int cnt = min(cnt_raw, p.n);
  if (r >= cnt) {
    for (int i = threadIdx.x * 4; i < nrows * P * C; i += blockDim.x * 4) st4(orow + i, make_float4(0, 0, 0, 0));
    return;
  }

  const int lvl = min(max(lvl_raw, 0), p.num_levels - 1);
  const FT* __restrict__ feat = reinterpret_cast<const FT*>(p.data[img][lvl]);
  const int H = p.H[lvl], W = p.W[lvl];
  constexpr int S = (POOL == ODET_ROI_POOL_NONE) ? 1 : 2;
  const int crop = P * S;

  // normalised box (y1,x1,y2,x2) exactly as the reference builds it
  float y1n, x1n, y2n, x2n;
  int Hs = H, Ws = W;     // dims of the map crop_and_resize samples (padded for TP_ALIGN)
  if (NORM == ODET_ROI_NORM_IMAGE) {
    y1n = roi.y / p.image_h; x1n = roi.x / p.image_w;            // roi_pooling.py:30-35
    y2n = roi.w / p.image_h; x2n = roi.z / p.image_w;
  } else if (NORM == ODET_ROI_NORM_STRIDE) {
    const float st = p.stride[lvl];
    const float hm = (float)(H - 1), wm = (float)(W - 1);
    y1n = (roi.y / st) / hm; x1n = (roi.x / st) / wm;            // roi_pooling.py:64,69-74
    y2n = (roi.w / st) / hm; x2n = (roi.z / st) / wm;
  } else {
    const float st = p.stride[lvl];
    const float off = PAD ? 1.0f : 0.0f;
    if (PAD) { Hs = H + 2; Ws = W + 2; }                         // roi_pooling.py:100
    float x0 = roi.x / st, y0 = roi.y / st;                      // :175
    float x1 = roi.z / st, y1 = roi.w / st;
    if (PAD) { x0 = x0 + off; y0 = y0 + off; x1 = x1 + off; y1 = y1 + off; }   // :101
    const float cs = (float)crop;
    const float sw = (x1 - x0) / cs, sh = (y1 - y0) / cs;        // :120-121
    const float imh = (float)(Hs - 1), imw = (float)(Ws - 1);
    x1n = (x0 + sw / 2.0f - 0.5f) / imw;                         // :124
    y1n = (y0 + sh / 2.0f - 0.5f) / imh;                         // :125
    const float nw = sw * (float)(crop - 1) / imw;               // :127
    const float nh = sh * (float)(crop - 1) / imh;               // :128
    y2n = y1n + nh; x2n = x1n + nw;                              // :130
  }
  const Axis ay = make_axis(y1n, y2n, Hs, crop);
  const Axis ax = make_axis(x1n, x2n, Ws, crop);
  const int Hdim = PAD ? H : Hs, Wdim = PAD ? W : Ws;

  if (!STAGE && !PAD && S == 2 && p.use_desc) {
    // whole RoI per workgroup, per-bin descriptors built lane-parallel (P*P <= 64 checked on the host)
    roi_bins_desc<POOL, FT>(feat, W, C, P, crop, ay, ax, orow);
    return;
  }
  if constexpr (!std::is_same<FT, float>::value) {
    return;     // float16 maps are only served by the descriptor form (checked on the host)
  } else {

  // ALL sample rows and columns, one per lane
  const bool lane_taps = crop <= 64;
  const Tap txl = make_tap<PAD>(ax, min(lane, crop - 1), crop, Wdim);
  const Tap tyl = make_tap<PAD>(ay, min(lane, crop - 1), crop, Hdim);

  // bounding tile of the cells this row taps; staged when it is small (only for the un-padded modes:
  // the tensorpack modes read through the clamped indices straight from the map)
  bool staged = false;
  int r0 = 0, c0 = 0, ncols = 0, trows = 0;
  if (STAGE && !PAD && nrows == 1 && lane_taps) {
    int rmin = 0x7fffffff, rmax = -1;
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const int yy = row0 * S + s;
      if (rl_i(tyl.ok ? 1 : 0, yy)) { rmin = min(rmin, rl_i(tyl.lo, yy)); rmax = max(rmax, rl_i(tyl.hi, yy)); }
    }
    // in-bounds sample columns: coordinates are monotone in the sample index
    int cmin = 0x7fffffff, cmax = -1;
    {
      const float first = axis_coord(ax, 0, crop), last = axis_coord(ax, crop - 1, crop);
      const float lo = fminf(first, last), hi = fmaxf(first, last);
      // any sample inside [0, limit] lies in [max(lo,0), min(hi,limit)]; the bounding cells of that
      // interval contain every tapped column (a superset is fine: it only stages a few more cells)
      const float a = fmaxf(lo, 0.0f), b = fminf(hi, ax.limit);
      if (a <= b) { cmin = (int)floorf(a); cmax = (int)ceilf(b); }
    }
    if (rmax >= 0 && cmax >= 0) {
      trows = rmax - rmin + 1;
      ncols = cmax - cmin + 1;
      r0 = rmin; c0 = cmin;
      staged = (size_t)trows * ncols * C * 4 <= ROI_LDS_BYTES && trows * ncols < 2 * crop * S;
    }
  }
  if (STAGE && staged) {
    // one coalesced pass: cell = wave-strided, channels = lanes x float4
    const int cells = trows * ncols;
    for (int cell = w; cell < cells; cell += 4) {
      const int rr = cell / ncols, cc = cell - rr * ncols;
      const float* src = feat + ((size_t)(r0 + rr) * W + (c0 + cc)) * C;
      float* dst = tile + (size_t)cell * C;
      for (int c = lane * 4; c < C; c += 256)
        *reinterpret_cast<float4*>(dst + c) = *reinterpret_cast<const float4*>(src + c);
    }
    __syncthreads();
    roi_bins<POOL, PAD, true>(tile, ncols, r0, c0, C, P, crop, Hdim, Wdim, ay, ax, tyl, txl, row0, nrows, orow);
  } else {
    if (lane_taps) roi_bins<POOL, PAD, true>(feat, W, 0, 0, C, P, crop, Hdim, Wdim, ay, ax, tyl, txl, row0, nrows, orow);
    else roi_bins<POOL, PAD, false>(feat, W, 0, 0, C, P, crop, Hdim, Wdim, ay, ax, tyl, txl, row0, nrows, orow);
  }
  }   // FT == float
}

// The float16 instantiations live in their own translation unit (roi_half.hip = this file with ODET_ROI_HALF_TU):
// roi.hip is compiled with -fno-slp-vectorize (hipcc's SLP pass packs the float32 lerps into v_pk_*_f32, which
// issue slower than the scalar forms: +1.2 % for the float32 kernel), but the same flag costs the float16 kernel
// its packed conversions (357 instead of 237 us for the 8-image launch), so that one is built with SLP.
#define ODET_ROI_HALF_KERNELS(X)                                                                               \
  X(ODET_ROI_POOL_NONE, ODET_ROI_NORM_STRIDE) X(ODET_ROI_POOL_NONE, ODET_ROI_NORM_IMAGE)                       \
  X(ODET_ROI_POOL_NONE, ODET_ROI_NORM_TP_ALIGN) X(ODET_ROI_POOL_NONE, ODET_ROI_NORM_TP_ALIGN_NOPAD)            \
  X(ODET_ROI_POOL_MAX2, ODET_ROI_NORM_STRIDE) X(ODET_ROI_POOL_MAX2, ODET_ROI_NORM_IMAGE)                       \
  X(ODET_ROI_POOL_MAX2, ODET_ROI_NORM_TP_ALIGN) X(ODET_ROI_POOL_MAX2, ODET_ROI_NORM_TP_ALIGN_NOPAD)            \
  X(ODET_ROI_POOL_AVG2, ODET_ROI_NORM_STRIDE) X(ODET_ROI_POOL_AVG2, ODET_ROI_NORM_IMAGE)                       \
  X(ODET_ROI_POOL_AVG2, ODET_ROI_NORM_TP_ALIGN) X(ODET_ROI_POOL_AVG2, ODET_ROI_NORM_TP_ALIGN_NOPAD)
#ifdef ODET_ROI_HALF_TU
#define ODET_ROI_X(POOL, NORM) template __global__ void k_roi_pool<POOL, NORM, false, __half>(RoiParams);
ODET_ROI_HALF_KERNELS(ODET_ROI_X)
#undef ODET_ROI_X
#else
#define ODET_ROI_X(POOL, NORM) extern template __global__ void k_roi_pool<POOL, NORM, false, __half>(RoiParams);
ODET_ROI_HALF_KERNELS(ODET_ROI_X)
#undef ODET_ROI_X

// A/B switches for profiling, read once (function-local static: thread-safe initialisation).
//   ODET_ROI_STAGE=1   the LDS-staged tile path.  OFF by default: every bilinear tap still has to be read
//                      once from LDS (128 B/clk/CU, only 2x the vector L1's 64 B/clk/CU) behind a
//                      load -> ds_write -> barrier chain, and it measured 25-30 % slower than deduplicating
//                      the shared cells in registers (RoI kernel)
//   ODET_ROI_DESC=0    per-row / per-bin tap computation instead of the lane-parallel bin descriptors
//   ODET_ROI_ROWS=k    output rows per workgroup of the non-descriptor form
//   ODET_ROI_THREADS   256 | 512 threads per workgroup of the descriptor form
struct RoiEnv {
  int stage, desc, rows, threads, xcd_images;
  RoiEnv() {
    const char* e = getenv("ODET_ROI_STAGE");
    stage = (e && e[0] == '1') ? 1 : 0;
    e = getenv("ODET_ROI_DESC");
    desc = e ? atoi(e) : 1;
    e = getenv("ODET_ROI_ROWS");
    rows = e ? atoi(e) : 0;
    e = getenv("ODET_ROI_THREADS");
    threads = e ? atoi(e) : 512;
    e = getenv("ODET_ROI_XCD_IMAGES");
    xcd_images = e ? atoi(e) : 1;
  }
};
static const RoiEnv& roi_env() {
  static const RoiEnv env;
  return env;
}
static bool roi_stage_enabled() { return roi_env().stage != 0; }

template <int POOL, int NORM>
static void roi_launch(dim3 grid, int threads, hipStream_t st, const RoiParams& p, RoiEvents ev) {
  if (p.f16)
    hipExtLaunchKernelGGL(HIP_KERNEL_NAME(k_roi_pool<POOL, NORM, false, __half>), grid, dim3(threads), 0, st, ev.start,
                          ev.stop, 0, p);
  else if (NORM != ODET_ROI_NORM_TP_ALIGN && roi_stage_enabled())
    hipExtLaunchKernelGGL(HIP_KERNEL_NAME(k_roi_pool<POOL, NORM, true, float>), grid, dim3(threads), ROI_LDS_BYTES, st,
                          ev.start, ev.stop, 0, p);
  else
    hipExtLaunchKernelGGL(HIP_KERNEL_NAME(k_roi_pool<POOL, NORM, false, float>), grid, dim3(threads), 0, st, ev.start,
                          ev.stop, 0, p);
}

template <int POOL>
static void roi_launch_norm(int norm_mode, dim3 grid, int threads, hipStream_t st, const RoiParams& p, RoiEvents ev) {
  switch (norm_mode) {
    case ODET_ROI_NORM_STRIDE: roi_launch<POOL, ODET_ROI_NORM_STRIDE>(grid, threads, st, p, ev); break;
    case ODET_ROI_NORM_IMAGE: roi_launch<POOL, ODET_ROI_NORM_IMAGE>(grid, threads, st, p, ev); break;
    case ODET_ROI_NORM_TP_ALIGN: roi_launch<POOL, ODET_ROI_NORM_TP_ALIGN>(grid, threads, st, p, ev); break;
    default: roi_launch<POOL, ODET_ROI_NORM_TP_ALIGN_NOPAD>(grid, threads, st, p, ev); break;
  }
}

// B images in one launch (blockIdx.y = image); all images share shapes and parameters
int odet_roi_pool_batch(const RoiImageIO* io, int B, int num_levels, int C, int n, int norm_mode, int image_h,
                        int image_w, int pool_size, int pool_mode, hipStream_t st, RoiEvents ev, int f16) {
  ODET_REQUIRE(n >= 0, "odet_roi_pool: negative n");
  if (n == 0) return ODET_OK;
  ODET_REQUIRE(io && B >= 1 && B <= ODET_MAX_BATCH, "odet_roi_pool: bad batch");
  ODET_REQUIRE(num_levels > 0 && num_levels <= ODET_MAX_LEVELS, "odet_roi_pool: num_levels %d out of range", num_levels);
  ODET_REQUIRE(C > 0 && (C & 3) == 0, "odet_roi_pool: C must be a positive multiple of 4 (got %d)", C);
  ODET_REQUIRE(pool_size > 0 && pool_size <= 64, "odet_roi_pool: pool_size out of range");
  ODET_REQUIRE(norm_mode >= 0 && norm_mode <= 3, "odet_roi_pool: unknown norm_mode %d", norm_mode);
  ODET_REQUIRE(pool_mode >= 0 && pool_mode <= 2, "odet_roi_pool: unknown pool_mode %d", pool_mode);
  if (norm_mode == ODET_ROI_NORM_IMAGE) ODET_REQUIRE(image_h > 0 && image_w > 0, "odet_roi_pool: bad image shape");
  RoiParams p;
  for (int i = 0; i < ODET_MAX_BATCH; ++i) {
    const RoiImageIO& a = io[i < B ? i : 0];
    ODET_REQUIRE(a.levels && a.rois && a.out, "odet_roi_pool: null pointer");
    ODET_REQUIRE(num_levels == 1 || a.roi_level, "odet_roi_pool: roi_level required with several levels");
    for (int l = 0; l < ODET_MAX_LEVELS; ++l) {
      const odet_level_t* L = &a.levels[l < num_levels ? l : 0];
      ODET_REQUIRE(L->data && L->H > 0 && L->W > 0, "odet_roi_pool: bad level %d", l);
      if (norm_mode != ODET_ROI_NORM_IMAGE) ODET_REQUIRE(L->stride > 0.0f, "odet_roi_pool: bad stride on level %d", l);
      if (i > 0 && i < B)
        ODET_REQUIRE(L->H == p.H[l] && L->W == p.W[l] && L->stride == p.stride[l],
                     "odet_roi_pool: the images of a batch must share the level shapes");
      p.data[i][l] = L->data;
      if (i == 0) { p.H[l] = L->H; p.W[l] = L->W; p.stride[l] = L->stride; }
    }
    p.rois.v[i] = (const float4*)a.rois; p.roi_level.v[i] = a.roi_level; p.count_dev.v[i] = a.count_dev;
    p.order.v[i] = a.order;
    p.out.v[i] = a.out;
  }
  p.num_levels = num_levels;
  p.C = C; p.n = n; p.norm_mode = norm_mode; p.P = pool_size; p.pool_mode = pool_mode;
  p.image_h = (float)image_h; p.image_w = (float)image_w;
  const int rows_env = roi_env().rows;
  // the row form (un-pooled 7x7 crops, tensorpack modes, LDS staging) works best with one output row
  // per workgroup (measured: 39 us vs 45 us for whole-RoI workgroups); the descriptor form below takes
  // the whole RoI
  p.rows_per_wg = rows_env > 0 ? std::min(rows_env, pool_size) : 1;
  const int desc_env = roi_env().desc, threads_env = roi_env().threads;
  const bool desc_ok = pool_mode != ODET_ROI_POOL_NONE && norm_mode != ODET_ROI_NORM_TP_ALIGN &&
                       pool_size * pool_size <= 64;
  p.f16 = f16 ? 1 : 0;
  if (f16 && !desc_ok)
    return odet_set_error(ODET_E_INVALID, "odet_roi_pool_f16: float16 maps need a pooled (max / avg), un-padded mode "
                          "with pool_size <= 8");
  p.use_desc = (desc_ok && (f16 || (desc_env && !roi_stage_enabled()))) ? 1 : 0;
  if (p.use_desc) p.rows_per_wg = pool_size;
  p.groups_per_roi = (pool_size + p.rows_per_wg - 1) / p.rows_per_wg;
  const int threads = p.use_desc ? ((threads_env == 256 || threads_env == 512) ? threads_env : 512) : 256;
  int64_t rows = (int64_t)n * p.groups_per_roi;
  ODET_REQUIRE(rows < (1ll << 30), "odet_roi_pool: too many workgroups");
  p.nblocks = (int)rows;
  p.xcd_images = ((B == 2 || B == 4 || B == 8) && roi_env().xcd_images) ? 1 : 0;
  p.xcds_per_img = p.xcd_images ? 8 / B : 8;
  p.blocks_per_xcd = (p.nblocks + p.xcds_per_img - 1) / p.xcds_per_img;   // per XCD of an image
  dim3 grid(p.blocks_per_xcd * 8, p.xcd_images ? 1 : B);
  if (pool_mode == ODET_ROI_POOL_NONE) roi_launch_norm<ODET_ROI_POOL_NONE>(norm_mode, grid, threads, st, p, ev);
  else if (pool_mode == ODET_ROI_POOL_MAX2) roi_launch_norm<ODET_ROI_POOL_MAX2>(norm_mode, grid, threads, st, p, ev);
  else roi_launch_norm<ODET_ROI_POOL_AVG2>(norm_mode, grid, threads, st, p, ev);
  ODET_LAUNCH_CHECK();
  return ODET_OK;
}

// ---- spatial processing order -------------------------------------------------------------------
// Output row r always holds RoI r; only the ORDER in which workgroups pick RoIs changes: sorted by
// (level, y centre, x centre) so that the RoIs an XCD processes (a contiguous chunk of the order) tap one
// band of one pyramid level -- fewer lines fetched by several XCDs, and a sliding working set in each L2.
struct RoiOrderParams {
  PerImg<const float4*> rois;
  PerImg<const int32_t*> roi_level;
  PerImg<const int32_t*> count_dev;
  PerImg<int32_t*> order;
  int n, P2;
  float inv_h, inv_w;
};

__global__ void __launch_bounds__(1024) k_roi_order(RoiOrderParams p) {
  extern __shared__ __align__(16) unsigned long long okeys[];   // [max(P2, 2048)]
  const int img = blockIdx.y;
  const float4* __restrict__ rois = p.rois.v[img];
  const int32_t* __restrict__ lvl = p.roi_level.v[img];
  const int32_t* __restrict__ cd = p.count_dev.v[img];
  int32_t* __restrict__ order = p.order.v[img];
  const int cnt = cd ? min(*cd, p.n) : p.n;
  const int tid = threadIdx.x;
  auto make_key = [&](int r) -> unsigned long long {
    if (r >= p.n) return ~0ull;
    if (r >= cnt) return (0xFFFFFFFEull << 32) | (unsigned)r;          // padded rows last (they are zero-filled)
    const float4 b = rois[r];
    const int l = lvl ? min(max(lvl[r], 0), 7) : 0;
    const int qy = min(max((int)((b.y + b.w) * 0.5f * p.inv_h * 4096.0f), 0), 4095);
    const int qx = min(max((int)((b.x + b.z) * 0.5f * p.inv_w * 4096.0f), 0), 4095);
    return ((unsigned long long)((l << 24) | (qy << 12) | qx) << 32) | (unsigned)r;
  };
  if (p.P2 <= 1024) {
    unsigned long long k = make_key(tid);
    k = bitonic_sort_1024_reg(k, okeys);
    if (tid < p.n) order[tid] = (int32_t)(k & 0xFFFFFFFFull);
  } else {
    for (int i = tid; i < p.P2; i += 1024) okeys[i] = make_key(i);
    __syncthreads();
    bitonic_sort_u64(okeys, p.P2, 1024);
    for (int i = tid; i < p.n; i += 1024) order[i] = (int32_t)(okeys[i] & 0xFFFFFFFFull);
  }
}

int odet_roi_order_batch(const RoiOrderIO* io, int B, int n, int image_h, int image_w, hipStream_t st) {
  ODET_REQUIRE(io && B >= 1 && B <= ODET_MAX_BATCH, "odet_roi_order: bad batch");
  ODET_REQUIRE(n >= 0 && n <= 8192, "odet_roi_order: n %d out of range (<= 8192)", n);
  ODET_REQUIRE(image_h > 0 && image_w > 0, "odet_roi_order: bad image shape");
  if (n == 0) return ODET_OK;
  RoiOrderParams p;
  for (int i = 0; i < ODET_MAX_BATCH; ++i) {
    const RoiOrderIO& a = io[i < B ? i : 0];
    ODET_REQUIRE(a.rois && a.order, "odet_roi_order: null pointer");
    p.rois.v[i] = (const float4*)a.rois; p.roi_level.v[i] = a.roi_level; p.count_dev.v[i] = a.count_dev;
    p.order.v[i] = a.order;
  }
  p.n = n;
  p.P2 = 2;
  while (p.P2 < n) p.P2 <<= 1;
  p.inv_h = 1.0f / (float)image_h; p.inv_w = 1.0f / (float)image_w;
  const size_t lds = (size_t)(p.P2 > 2048 ? p.P2 : 2048) * 8;
  static std::once_flag once;       // (executor threads may arrive here together)
  static hipError_t once_rc = hipSuccess;
  std::call_once(once, [] {
    once_rc = hipFuncSetAttribute((const void*)k_roi_order, hipFuncAttributeMaxDynamicSharedMemorySize, 8192 * 8);
  });
  ODET_HIP(once_rc);
  hipLaunchKernelGGL(k_roi_order, dim3(1, B), dim3(1024), lds, st, p);
  ODET_LAUNCH_CHECK();
  return ODET_OK;
}

extern "C" int odet_roi_order(const float* rois, const int32_t* roi_level, int n, const int32_t* count_dev, int image_h,
                              int image_w, int32_t* out_order, odet_stream_t stream) {
  RoiOrderIO io{rois, roi_level, count_dev, out_order};
  return odet_roi_order_batch(&io, 1, n, image_h, image_w, (hipStream_t)stream);
}

static int roi_pool_impl(const odet_level_t* levels, int num_levels, int C, const float* rois,
                         const int32_t* roi_level, int n, const int32_t* count_dev, const int32_t* order,
                         int norm_mode, int image_h, int image_w, int pool_size, int pool_mode, float* out,
                         odet_stream_t stream, RoiEvents ev) {
  RoiImageIO io{levels, rois, roi_level, count_dev, order, out};
  if (n > 0) ODET_REQUIRE(levels && rois && out, "odet_roi_pool: null pointer");
  return odet_roi_pool_batch(&io, 1, num_levels, C, n, norm_mode, image_h, image_w, pool_size, pool_mode,
                             (hipStream_t)stream, ev, 0);
}

extern "C" int odet_roi_pool_f16(const odet_level_t* levels, int num_levels, int C, const float* rois,
                                 const int32_t* roi_level, int n, const int32_t* count_dev, const int32_t* order,
                                 int norm_mode, int image_h, int image_w, int pool_size, int pool_mode, void* out,
                                 odet_stream_t stream) {
  RoiImageIO io{levels, rois, roi_level, count_dev, order, (float*)out};
  if (n > 0) ODET_REQUIRE(levels && rois && out, "odet_roi_pool_f16: null pointer");
  return odet_roi_pool_batch(&io, 1, num_levels, C, n, norm_mode, image_h, image_w, pool_size, pool_mode,
                             (hipStream_t)stream, RoiEvents{nullptr, nullptr}, 1);
}

extern "C" int odet_roi_pool_f16_timed(const odet_level_t* levels, int num_levels, int C, const float* rois,
                                       const int32_t* roi_level, int n, const int32_t* count_dev,
                                       const int32_t* order, int norm_mode, int image_h, int image_w, int pool_size,
                                       int pool_mode, void* out, odet_stream_t stream, void* start_event,
                                       void* stop_event) {
  RoiImageIO io{levels, rois, roi_level, count_dev, order, (float*)out};
  if (n > 0) ODET_REQUIRE(levels && rois && out, "odet_roi_pool_f16_timed: null pointer");
  return odet_roi_pool_batch(&io, 1, num_levels, C, n, norm_mode, image_h, image_w, pool_size, pool_mode,
                             (hipStream_t)stream, RoiEvents{(hipEvent_t)start_event, (hipEvent_t)stop_event}, 1);
}

extern "C" int odet_roi_pool(const odet_level_t* levels, int num_levels, int C, const float* rois,
                             const int32_t* roi_level, int n, const int32_t* count_dev, int norm_mode,
                             int image_h, int image_w, int pool_size, int pool_mode, float* out,
                             odet_stream_t stream) {
  return roi_pool_impl(levels, num_levels, C, rois, roi_level, n, count_dev, nullptr, norm_mode, image_h, image_w,
                       pool_size, pool_mode, out, stream, RoiEvents{nullptr, nullptr});
}

extern "C" int odet_roi_pool_ordered(const odet_level_t* levels, int num_levels, int C, const float* rois,
                                     const int32_t* roi_level, int n, const int32_t* count_dev,
                                     const int32_t* order, int norm_mode, int image_h, int image_w, int pool_size,
                                     int pool_mode, float* out, odet_stream_t stream, void* start_event,
                                     void* stop_event) {
  return roi_pool_impl(levels, num_levels, C, rois, roi_level, n, count_dev, order, norm_mode, image_h, image_w,
                       pool_size, pool_mode, out, stream, RoiEvents{(hipEvent_t)start_event, (hipEvent_t)stop_event});
}

extern "C" int odet_roi_pool_timed(const odet_level_t* levels, int num_levels, int C, const float* rois,
                                   const int32_t* roi_level, int n, const int32_t* count_dev, int norm_mode,
                                   int image_h, int image_w, int pool_size, int pool_mode, float* out,
                                   odet_stream_t stream, void* start_event, void* stop_event) {
  ODET_REQUIRE(start_event && stop_event, "odet_roi_pool_timed: null event");
  return roi_pool_impl(levels, num_levels, C, rois, roi_level, n, count_dev, nullptr, norm_mode, image_h, image_w,
                       pool_size, pool_mode, out, stream, RoiEvents{(hipEvent_t)start_event, (hipEvent_t)stop_event});
}

extern "C" int odet_prof_event_create(void** ev) {
  ODET_REQUIRE(ev, "odet_prof_event_create: null pointer");
  hipEvent_t e;
  ODET_HIP(hipEventCreate(&e));
  *ev = (void*)e;
  return ODET_OK;
}

extern "C" int odet_prof_event_destroy(void* ev) {
  if (ev) ODET_HIP(hipEventDestroy((hipEvent_t)ev));
  return ODET_OK;
}

extern "C" int odet_prof_event_elapsed_ms(void* start, void* stop, float* ms) {
  ODET_REQUIRE(start && stop && ms, "odet_prof_event_elapsed_ms: null pointer");
  ODET_HIP(hipEventSynchronize((hipEvent_t)stop));
  ODET_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
  return ODET_OK;
}
#endif   // !ODET_ROI_HALF_TU

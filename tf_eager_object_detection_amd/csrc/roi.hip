// RoI feature extraction: tf.image.crop_and_resize (bilinear, extrapolation 0) fused with the
// reference's 2x2 max / avg pooling, over all pyramid levels (and up to 8 images) in one launch.
//
//   model/roi_pooling.py:45-90   RoiPoolingCropAndResize   (NORM_STRIDE, POOL_MAX2 | POOL_NONE)
//   model/roi_pooling.py:8-42    RoiPoolingCropAndResize2  (NORM_IMAGE,  POOL_MAX2)   <- FPN
//   model/roi_pooling.py:93-177  crop_and_resize/roi_align/RoiPoolingRoiAlign
//                                                          (NORM_TP_ALIGN[_NOPAD], POOL_AVG2 | POOL_NONE)
//
// The reference materialises the [R,2P,2P,C] crops (200 MB at R = 1000, C = 256) and pools them in a
// second op; here the pooled [R,P,P,C] features are the only thing written.
//
// Work split: ONE WAVE = ONE OUTPUT ROW of one RoI (P bins), a lane = 4 consecutive channels of a cell
// (NHWC: a cell's C channels are contiguous, 64 lanes x 16 B = 1 KiB coalesced per cell and wave).  A
// workgroup is the P waves of one RoI (they tap the same cells, so they share the CU's L1).
//
//   * Everything about the sample ROWS of the wave's output row (the tapped cell rows, their lerp
//     weights, how the two sample rows of a bin share cell rows) is wave-uniform and loop-invariant: it
//     lives in scalar registers.  The sample COLUMNS are computed once, lane px = bin column px, and a bin
//     fetches its column descriptor with four v_readlane.
//   * Cells are read with buffer loads whose address is split the way the data is: the per-lane part
//     (lane * 16 B, + a column step) never changes, the per-cell part is a scalar byte offset (soffset) --
//     a bin costs a handful of scalar adds and no vector address arithmetic.
//   * Cells tapped by both samples of a bin along an axis are DEDUPLICATED IN REGISTERS: along an axis
//     the two samples tap cells (lo0, lo0+1) and (lo1, lo1+1); D = lo1 - lo0 in {0, 1} means they share
//     cells, so only 2 + D distinct rows / columns are loaded (4, 6 or 9 cells instead of 16 taps -- every
//     RoI whose sample spacing is below one cell); D = 2 is the general form (lo0, hi0, lo1, hi1 as they
//     are).  Same values, same lerp arithmetic -> bit-identical to the 16-tap form.  The row class is fixed
//     per wave (three loop variants), the column class is switched per bin.
//   * Lerps in the exact TF operation order (no FMA: the library is built with -ffp-contract=off), the
//     2x2 max / avg reduced in registers -- max is exact and avg uses the same row-major sum, so results
//     are bit-identical to the un-fused form.  Pooled features are written once with non-temporal stores.
//   * XCD-aware placement: consecutive (spatially ordered, odet_roi_order) RoIs go to the same XCD, and in
//     a batch of 2 / 4 / 8 images every image gets 4 / 2 / 1 XCDs of its own, so an XCD's L2 holds one
//     neighbourhood of one pyramid level of one image.
//
// Measured alternatives that lost (LDS-staged RoI tiles, LDS-DMA prefetch ring, packed float32 lerps,
// row sharing between bins) are recorded in DESIGN.md section 3.2 and tools/exp/*.patch.
#include <hip/hip_ext.h>
#include <hip/hip_fp16.h>

#include <mutex>
#include <type_traits>

#include "odet_internal.h"

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
  int C, n, P, num_levels;
  float image_h, image_w;
  int nblocks;        // logical workgroups per image (before padding the grid to a multiple of 8)
  int blocks_per_xcd;
  int waves;          // waves (= output rows) per workgroup; == P: a workgroup is one RoI
  int xcd_images;     // 1: the image is derived from the XCD slot (batch of 2 / 4 / 8), 0: blockIdx.y
  int xcds_per_img;   // XCDs that serve one image (8 / batch)
  int slices;         // > 1: C / 256 workgroups per RoI, one 256-channel slice each
  int rois_per_xcd;   // slices > 1: an XCD walks its RoIs once per slice, SLICE-MAJOR (all of slice 0, then slice 1, ...)
  int roi_groups;     // > 0 (slices > 1 and the image's XCDs are a multiple of the slices): an XCD serves ONE slice for one of
                      // roi_groups contiguous parts of the processing order (rois_per_xcd RoIs each)
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

struct Tap {      // one sample along one axis
  bool ok;        // TF: not extrapolated (0 <= in <= dim-1; NaN fails)
  int lo, hi;     // floor / ceil cell (after the SYMMETRIC-pad remap for the padded tensorpack mode)
  float lerp;
};

template <bool PAD>
__device__ __forceinline__ Tap make_tap(const Axis& a, int i, int crop, int dim) {
  Tap t;
  const float in = (crop > 1) ? a.start + (float)i * a.scale : a.single;
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
  if (!t.ok) { lo = 0; hi = 0; }     // never an address
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

// ---- cell I/O through buffer descriptors -------------------------------------------------------------------
// address = descriptor base + soffset (scalar, bytes) + voffset (per lane, bytes).  float32 maps as they are,
// float16 maps (BASELINE config 5) widened to float32 for the lerps and rounded to nearest-even on the way out.
typedef uint32_t u4v __attribute__((ext_vector_type(4)));
typedef uint32_t u2v __attribute__((ext_vector_type(2)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

__device__ __forceinline__ rsrc_t make_rsrc(const void* base, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

template <typename FT> struct Cell;
// t = l + (r - l) * w: the horizontal half of TF's bilinear sample (crop_and_resize_op.cc), one IEEE operation each
__device__ __forceinline__ float lerp1(float l, float r, float w) { return l + (r - l) * w; }

template <> struct Cell<float> {
  static constexpr uint32_t LANE_BYTES = 16;
  typedef u4v raw_t;              // a cell's 4 channels of this lane as they come out of memory
  static __device__ __forceinline__ raw_t load_raw(rsrc_t r, uint32_t voff, uint32_t soff) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
  }
  static __device__ __forceinline__ float4 widen(raw_t u) {
    return make_float4(__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w));
  }
  static __device__ __forceinline__ float4 load(rsrc_t r, uint32_t voff, uint32_t soff) { return widen(load_raw(r, voff, soff)); }
  // l + (r - l) * w per channel
  static __device__ __forceinline__ float4 hlerp(raw_t l, raw_t r, float w) {
    const float4 a = widen(l), b = widen(r);
    return make_float4(lerp1(a.x, b.x, w), lerp1(a.y, b.y, w), lerp1(a.z, b.z, w), lerp1(a.w, b.w, w));
  }
  // Pooled features are written once and not read again by this path: non-temporal stores keep the 50 MB per
  // image out of L2 / Infinity Cache, where they would evict the feature-map lines neighbouring RoIs share.
  static __device__ __forceinline__ void store(rsrc_t r, uint32_t voff, uint32_t soff, float4 v) {
    const u4v u = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
    __builtin_amdgcn_raw_buffer_store_b128(u, r, (int)voff, (int)soff, /*nt*/ 2);
  }
};
// float16 maps: the horizontal lerp of one channel straight from the packed halves, WITHOUT separate conversions --
// v_fma_mix_f32 reads a float16 operand (low / high half of a register by op_sel) widened exactly to float32 inside the
// instruction, and  a * 1.0 + (-b)  is the correctly rounded float32 difference a - b,  a * 1.0 + m  the correctly
// rounded sum: the same three IEEE operations (sub, mul, add) on the same float32 values as convert-then-lerp, bit for
// bit, minus the conversions (36 of ~170 vector instructions of a 3 x 3-cell bin: the float16 kernel is bound by its
// vector-instruction issue).
template <int HI>
__device__ __forceinline__ float lerp1_f16(uint32_t l, uint32_t r, float w) {
  float d, o;
  if (HI) asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(r), "v"(l));
  else asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(r), "v"(l));
  const float m = d * w;
  if (HI) asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(o) : "v"(l), "v"(m));
  else asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(o) : "v"(l), "v"(m));
  return o;
}

template <> struct Cell<__half> {
  static constexpr uint32_t LANE_BYTES = 8;
  typedef u2v raw_t;
  static __device__ __forceinline__ raw_t load_raw(rsrc_t r, uint32_t voff, uint32_t soff) {
    return __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, 0);
  }
  static __device__ __forceinline__ float4 widen(raw_t u) {
    const uint32_t ux = u.x, uy = u.y;
    const __half2 a = *reinterpret_cast<const __half2*>(&ux), b = *reinterpret_cast<const __half2*>(&uy);
    const float2 fa = __half22float2(a), fb = __half22float2(b);
    return make_float4(fa.x, fa.y, fb.x, fb.y);
  }
  static __device__ __forceinline__ float4 load(rsrc_t r, uint32_t voff, uint32_t soff) { return widen(load_raw(r, voff, soff)); }
  static __device__ __forceinline__ float4 hlerp(raw_t l, raw_t r, float w) {
    return make_float4(lerp1_f16<0>(l.x, r.x, w), lerp1_f16<1>(l.x, r.x, w), lerp1_f16<0>(l.y, r.y, w), lerp1_f16<1>(l.y, r.y, w));
  }
  static __device__ __forceinline__ void store(rsrc_t r, uint32_t voff, uint32_t soff, float4 v) {
    const __half2 a = __floats2half2_rn(v.x, v.y), b = __floats2half2_rn(v.z, v.w);
    u2v u;
    u.x = *reinterpret_cast<const uint32_t*>(&a);
    u.y = *reinterpret_cast<const uint32_t*>(&b);
    __builtin_amdgcn_raw_buffer_store_b64(u, r, (int)voff, (int)soff, 0);   // (float16 features are read back by the RoI head)
  }
};

__device__ __forceinline__ int rl_i(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ uint32_t rl_u(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }
__device__ __forceinline__ float rl_f(float v, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
__device__ __forceinline__ int rfl_i(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint32_t rfl_u(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ float rfl_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}

// sharing class of the two samples of a bin along one axis: 0 / 1 = they share cells (lo1 - lo0), 2 = general
__device__ __forceinline__ int share_class(const Tap& t0, const Tap& t1) {
  const int d = t1.lo - t0.lo;
  return (t0.hi == t0.lo + 1 && t1.hi == t1.lo + 1 && (d == 0 || d == 1)) ? d : 2;
}

// What a wave knows about its output row (scalar registers): byte offsets of the four tapped cell rows
// (lo0, hi0, lo1, hi1; for the sharing classes row[i] = row[0] + i * row pitch), the y lerp weights, the maps'
// descriptor, and the per-lane byte offset.
struct RowCtx {
  rsrc_t feat;
  uint32_t row[4];
  float yw[2];
  uint32_t vlane;     // lane * bytes per lane (+ channel chunk)
  uint32_t cellB;     // bytes per cell = C * sizeof(FT)
};

// One bin whose 2 x 2 samples are all inside the map.  DY / DX: sharing class per axis.  c0: byte offset of the
// bin's first cell column; crel[k]: byte offsets of the columns (lo0, hi0, lo1, hi1) relative to c0 (DX == 2).
template <int POOL, int DY, int DX, typename FT>
__device__ __forceinline__ float4 roi_bin(const RowCtx& rc, uint32_t c0, const uint32_t (&crel)[4], const float (&xw)[2]) {
  constexpr int NR = (DY == 2) ? 4 : 2 + DY, NC = (DX == 2) ? 4 : 2 + DX;
  uint32_t soff[NR], voff[NC];
#pragma unroll
  for (int i = 0; i < NR; ++i) soff[i] = rc.row[i] + c0;                              // scalar
#pragma unroll
  for (int j = 0; j < NC; ++j) voff[j] = rc.vlane + ((DX == 2) ? crel[j] : (uint32_t)j * rc.cellB);
  typename Cell<FT>::raw_t blk[NR][NC];
#pragma unroll
  for (int i = 0; i < NR; ++i) {
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      blk[i][j] = Cell<FT>::load_raw(rc.feat, voff[j], soff[i]);
    }
  }
  // TF's sample = top + (bottom - top) * y_lerp with top / bottom = left + (right - left) * x_lerp (crop_and_resize_op.cc):
  // the horizontal lerp of every tapped row for both sample columns (a row shared by the bin's two sample rows is
  // lerped once: same operands, same result), then the vertical ones
  float4 h[NR][2];
#pragma unroll
  for (int i = 0; i < NR; ++i) {
#pragma unroll
    for (int sx = 0; sx < 2; ++sx) {
      const int cl = (DX == 2) ? 2 * sx : sx * DX;
      h[i][sx] = Cell<FT>::hlerp(blk[i][cl], blk[i][cl + 1], xw[sx]);
    }
  }
  float4 v[2][2];
#pragma unroll
  for (int sy = 0; sy < 2; ++sy) {
#pragma unroll
    for (int sx = 0; sx < 2; ++sx) {
      const int rt = (DY == 2) ? 2 * sy : sy * DY;
      const float4 t = h[rt][sx], b = h[rt + 1][sx];
      const float yw = rc.yw[sy];
      v[sy][sx] = make_float4(lerp1(t.x, b.x, yw), lerp1(t.y, b.y, yw), lerp1(t.z, b.z, yw), lerp1(t.w, b.w, yw));
    }
  }
  return pool4<POOL>(v);
}

// The same bin (column classes 0 / 1: NC = 2 + DX consecutive columns from c0) with its first M columns taken from the
// registers of the previous bin of the row instead of memory.  Neighbouring bins of an output row overlap whenever
// the samples are less than a cell apart (the usual case: median spacing 0.69 cells on the bench workload): the
// previous bin's last column is this bin's first (M = 1), or its last two are this bin's first two (M = 2; with
// NC = 2 the bin then loads nothing).  car[i][0..1]: the previous bin's last two columns (row i); on return this
// bin's.  The values and the lerps are the ones roi_bin computes: results are bit-identical.
template <int POOL, int DY, int DX, int M, typename FT>
__device__ __forceinline__ float4 roi_bin_carry(const RowCtx& rc, uint32_t c0, const float (&xw)[2],
                                                float4 (&car)[(DY == 2) ? 4 : 2 + DY][2]) {
  static_assert(DX == 0 || DX == 1, "consecutive columns only");
  constexpr int NR = (DY == 2) ? 4 : 2 + DY, NC = 2 + DX;
  uint32_t soff[NR], voff[NC];
#pragma unroll
  for (int i = 0; i < NR; ++i) soff[i] = rc.row[i] + c0;
#pragma unroll
  for (int j = 0; j < NC; ++j) voff[j] = rc.vlane + (uint32_t)j * rc.cellB;
  float4 blk[NR][NC];
#pragma unroll
  for (int i = 0; i < NR; ++i) {
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      if (j < M) blk[i][j] = car[i][2 - M + j];
      else blk[i][j] = Cell<FT>::load(rc.feat, voff[j], soff[i]);
    }
  }
  float4 v[2][2];
#pragma unroll
  for (int sy = 0; sy < 2; ++sy) {
#pragma unroll
    for (int sx = 0; sx < 2; ++sx) {
      const int rt = (DY == 2) ? 2 * sy : sy * DY, rb = rt + 1;
      const int cl = sx * DX, cr = cl + 1;
      v[sy][sx] = lerp_tap(blk[rt][cl], blk[rt][cr], blk[rb][cl], blk[rb][cr], xw[sx], rc.yw[sy]);
    }
  }
#pragma unroll
  for (int i = 0; i < NR; ++i) { car[i][0] = blk[i][NC - 2]; car[i][1] = blk[i][NC - 1]; }
  return pool4<POOL>(v);
}

// One 256-channel slice of an output row (a lane keeps ITS 4 channels of the carried columns): the bins of the row
// from left to right, each taking the columns it shares with its left neighbour from
// registers (roi_bin_carry).  7.9 -> ~5 cell loads per bin on the bench workload.
template <int POOL, int DY, typename FT>
__device__ __forceinline__ void roi_row_carry_pass(const RowCtx& rc, rsrc_t out, int P, uint32_t xcls_l, uint32_t c0_l,
                                                   const uint32_t (&crel_l)[4], float xw0_l, float xw1_l) {
  constexpr int NR = (DY == 2) ? 4 : 2 + DY;
  float4 car[NR][2];
#pragma unroll
  for (int i = 0; i < NR; ++i) car[i][0] = car[i][1] = make_float4(0, 0, 0, 0);
  uint32_t last = 0xFFFFFFFFu;        // byte offset (inside a cell row) of the carried last column; none yet
  for (int px = 0; px < P; ++px) {
    const int xc = rl_i((int)xcls_l, px);
    const uint32_t c0 = rl_u(c0_l, px);
    const float xw[2] = {rl_f(xw0_l, px), rl_f(xw1_l, px)};
    float4 o;
    if (xc == 2) {
      const uint32_t crel[4] = {0u, rl_u(crel_l[1], px), rl_u(crel_l[2], px), rl_u(crel_l[3], px)};
      o = roi_bin<POOL, DY, 2, FT>(rc, c0, crel, xw);
      last = 0xFFFFFFFFu;
    } else {
      const int m = (last == 0xFFFFFFFFu) ? 0 : (c0 == last ? 1 : (c0 + rc.cellB == last ? 2 : 0));
      switch (xc * 3 + m) {
        case 0: o = roi_bin_carry<POOL, DY, 0, 0, FT>(rc, c0, xw, car); break;
        case 1: o = roi_bin_carry<POOL, DY, 0, 1, FT>(rc, c0, xw, car); break;
        case 2: o = roi_bin_carry<POOL, DY, 0, 2, FT>(rc, c0, xw, car); break;
        case 3: o = roi_bin_carry<POOL, DY, 1, 0, FT>(rc, c0, xw, car); break;
        case 4: o = roi_bin_carry<POOL, DY, 1, 1, FT>(rc, c0, xw, car); break;
        default: o = roi_bin_carry<POOL, DY, 1, 2, FT>(rc, c0, xw, car); break;
      }
      last = c0 + (uint32_t)(1 + xc) * rc.cellB;
    }
    Cell<FT>::store(out, rc.vlane, (uint32_t)px * rc.cellB, o);
  }
}

// Maps of 256, 512, 1024, ... channels: one carrying pass per 256-channel slice (a lane's carried registers hold its 4
// channels of ONE slice), the slice in the per-lane byte offset of the loads and of the store.
template <int POOL, int DY, typename FT>
__device__ __forceinline__ void roi_row_carry(const RowCtx& rc0, rsrc_t out, int P, int C, int ch0, int ch1, uint32_t xcls_l, uint32_t c0_l,
                                              const uint32_t (&crel_l)[4], float xw0_l, float xw1_l) {
  RowCtx rc = rc0;
  for (int ch = ch0; ch < ch1; ch += 256) {
    rc.vlane = rc0.vlane + (uint32_t)ch * (Cell<FT>::LANE_BYTES / 4);
    roi_row_carry_pass<POOL, DY, FT>(rc, out, P, xcls_l, c0_l, crel_l, xw0_l, xw1_l);
  }
}

// The bins of one output row whose row class is DY, every sample of the row inside the map.  FULL: C is a
// multiple of 256 (no lane is ever idle).
template <int POOL, int DY, bool FULL, typename FT>
__device__ __forceinline__ void roi_row(const RowCtx& rc0, rsrc_t out, int P, int C, int ch0, int ch1, uint32_t xcls_l, uint32_t c0_l,
                                        const uint32_t (&crel_l)[4], float xw0_l, float xw1_l) {
  RowCtx rc = rc0;
  const int lane = threadIdx.x & 63;
  for (int px = 0; px < P; ++px) {
    const int xc = rl_i((int)xcls_l, px);
    const uint32_t c0 = rl_u(c0_l, px);
    const float xw[2] = {rl_f(xw0_l, px), rl_f(xw1_l, px)};
    uint32_t crel[4] = {0, 0, 0, 0};
    if (xc == 2) {
#pragma unroll
      for (int k = 1; k < 4; ++k) crel[k] = rl_u(crel_l[k], px);
    }
    const uint32_t so_out = (uint32_t)px * rc.cellB;
    for (int ch = ch0; ch < ch1; ch += 256) {
      rc.vlane = rc0.vlane + (uint32_t)ch * (Cell<FT>::LANE_BYTES / 4);
      if (FULL || lane * 4 + ch < C) {
        float4 o;
        if (xc == 0) o = roi_bin<POOL, DY, 0, FT>(rc, c0, crel, xw);
        else if (xc == 1) o = roi_bin<POOL, DY, 1, FT>(rc, c0, crel, xw);
        else o = roi_bin<POOL, DY, 2, FT>(rc, c0, crel, xw);
        Cell<FT>::store(out, rc.vlane, so_out, o);
      }
    }
  }
}

// An output row with extrapolated samples (TF: such a sample is 0): every sample guarded, 4 taps each; rc.row
// holds the four tapped rows (lo0, hi0, lo1, hi1) as they are.  Rare (boxes are clipped to the image).
template <int POOL, typename FT>
__device__ __forceinline__ void roi_row_guarded(const RowCtx& rc0, rsrc_t out, int P, int C, int ch0, int ch1, const uint32_t (&cabs_l)[4],
                                             uint32_t xok_l, float xw0_l, float xw1_l, uint32_t yok) {
  RowCtx rc = rc0;
  const int lane = threadIdx.x & 63;
  for (int px = 0; px < P; ++px) {
    uint32_t cabs[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) cabs[k] = rl_u(cabs_l[k], px);
    const uint32_t xok = rl_u(xok_l, px);
    const float xw[2] = {rl_f(xw0_l, px), rl_f(xw1_l, px)};
    const uint32_t so_out = (uint32_t)px * rc.cellB;
    for (int ch = ch0; ch < ch1; ch += 256) {
      rc.vlane = rc0.vlane + (uint32_t)ch * (Cell<FT>::LANE_BYTES / 4);
      if (lane * 4 + ch < C) {
        float4 v[2][2];
#pragma unroll
        for (int sy = 0; sy < 2; ++sy) {
#pragma unroll
          for (int sx = 0; sx < 2; ++sx) {
            float4 res = make_float4(0, 0, 0, 0);
            if (((yok >> sy) & 1) && ((xok >> sx) & 1)) {
              const float4 tl = Cell<FT>::load(rc.feat, rc.vlane, rc.row[2 * sy] + cabs[2 * sx]);
              const float4 tr = Cell<FT>::load(rc.feat, rc.vlane, rc.row[2 * sy] + cabs[2 * sx + 1]);
              const float4 bl = Cell<FT>::load(rc.feat, rc.vlane, rc.row[2 * sy + 1] + cabs[2 * sx]);
              const float4 br = Cell<FT>::load(rc.feat, rc.vlane, rc.row[2 * sy + 1] + cabs[2 * sx + 1]);
              res = lerp_tap(tl, tr, bl, br, xw[sx], rc.yw[sy]);
            }
            v[sy][sx] = res;
          }
        }
        Cell<FT>::store(out, rc.vlane, so_out, pool4<POOL>(v));
      }
    }
  }
}

// The un-pooled form (one sample per bin): 4 taps, guarded.
template <typename FT>
__device__ __forceinline__ void roi_row_single(const RowCtx& rc0, rsrc_t out, int P, int C, int ch0, int ch1, const uint32_t (&cabs_l)[4],
                                               uint32_t xok_l, float xw0_l, uint32_t yok) {
  RowCtx rc = rc0;
  const int lane = threadIdx.x & 63;
  for (int px = 0; px < P; ++px) {
    const uint32_t cl = rl_u(cabs_l[0], px), cr = rl_u(cabs_l[1], px);
    const uint32_t xok = rl_u(xok_l, px);
    const float xw = rl_f(xw0_l, px);
    const uint32_t so_out = (uint32_t)px * rc.cellB;
    const bool ok = (yok & 1) && (xok & 1);
    for (int ch = ch0; ch < ch1; ch += 256) {
      rc.vlane = rc0.vlane + (uint32_t)ch * (Cell<FT>::LANE_BYTES / 4);
      if (lane * 4 + ch < C) {
        float4 o = make_float4(0, 0, 0, 0);
        if (ok) {
          const auto tl = Cell<FT>::load_raw(rc.feat, rc.vlane, rc.row[0] + cl);
          const auto tr = Cell<FT>::load_raw(rc.feat, rc.vlane, rc.row[0] + cr);
          const auto bl = Cell<FT>::load_raw(rc.feat, rc.vlane, rc.row[1] + cl);
          const auto br = Cell<FT>::load_raw(rc.feat, rc.vlane, rc.row[1] + cr);
          const float4 t = Cell<FT>::hlerp(tl, tr, xw), b = Cell<FT>::hlerp(bl, br, xw);
          const float yw = rc.yw[0];
          o = make_float4(lerp1(t.x, b.x, yw), lerp1(t.y, b.y, yw), lerp1(t.z, b.z, yw), lerp1(t.w, b.w, yw));
        }
        Cell<FT>::store(out, rc.vlane, so_out, o);
      }
    }
  }
}

// TAG: 0 = the product launches; 1 = the SAME code under a second kernel name, launched when HIP events are
// attached to the dispatch (bench.py's roofline samples, timed alone on the GPU): a kernel-stats summary of a
// profiled run then lists those launches as a row of their own (k_roi_pool<.., 1>) instead of averaging them with
// the launches that share the chip with other streams' kernels.
template <int POOL, int NORM, typename FT, int TAG>
__global__ void __launch_bounds__(1024) k_roi_pool(RoiParams p) {
  constexpr bool PAD = (NORM == ODET_ROI_NORM_TP_ALIGN);
  constexpr int S = (POOL == ODET_ROI_POOL_NONE) ? 1 : 2;
  // XCD-aware remap: hardware deals workgroups round-robin over the 8 XCDs (each has its own L2).
  // A batch of 1 / 2 / 4 / 8 images gives every image 8 / 4 / 2 / 1 XCDs of its own, so that an XCD's L2
  // only ever holds lines of one image's maps; inside an image consecutive (spatially ordered) RoIs share an XCD.
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int img = p.xcd_images ? xcd / p.xcds_per_img : blockIdx.y;
  const int sub = p.xcd_images ? xcd - img * p.xcds_per_img : xcd;
  const int lb = sub * p.blocks_per_xcd + slot;
  if (slot >= p.blocks_per_xcd || lb >= p.nblocks) return;
  const int lane = threadIdx.x & 63;
  const int w = rfl_i(threadIdx.x >> 6);
  const int P = p.P, C = p.C;
  int ri, py;       // this wave: output row py of the ri-th RoI of the processing order
  int ch0 = 0, ch1 = C;     // ... and its channels
  if (p.slices > 1) {
    // maps of 512 / 1024 channels: a workgroup = one 256-channel slice of a RoI, and an XCD's workgroups are dealt
    // SLICE-MAJOR -- the first rois_per_xcd slots are slice 0 of its RoIs, the next ones slice 1, ... -- so that what
    // its L2 has to hold at a time is one 256-channel slice of the map (50 x 84 x 1 KB = 4.3 MB of the C4 map, not 17 MB):
    // the L2-miss stream of the 8-image C4 launch fell from 750 to ~200 MB (tools/pmc_roi_forms.sh)
    // With 8 or 4 XCDs for the image (launches of one or two images: the reference-surface layers) an XCD takes ONE slice
    // for a contiguous part of the (spatially ordered) RoIs instead: the 8 L2s then fetch the map once between them, not
    // once each -- big RoIs tap most of the map whatever their order (round 4: 1.84 x the distinct bytes from HBM).
    int sl;
    if (p.roi_groups > 0) {
      sl = sub % p.slices;
      ri = (sub / p.slices) * p.rois_per_xcd + slot;
      if (slot >= p.rois_per_xcd) return;
    } else {
      sl = slot / p.rois_per_xcd;
      ri = sub * p.rois_per_xcd + (slot - sl * p.rois_per_xcd);
    }
    py = w;
    ch0 = sl * 256; ch1 = ch0 + 256;
  } else if (p.waves == P) {
    ri = lb; py = w;
  } else {
    const int u = lb * p.waves + w;
    ri = u / P; py = u - ri * P;
  }
  if (ri >= p.n) return;
  const float4* __restrict__ rois = p.rois.v[img];
  const int32_t* __restrict__ roi_level = p.roi_level.v[img];
  const int32_t* __restrict__ count_dev = p.count_dev.v[img];
  const int32_t* __restrict__ order = p.order.v[img];
  const int r = order ? min(max(order[ri], 0), p.n - 1) : ri;
  // (independent loads: r < n always addresses valid rows)
  const int cnt_raw = count_dev ? *count_dev : p.n;
  const int lvl_raw = roi_level ? roi_level[r] : 0;
  const float4 roi = rois[r];
  const int cnt = min(cnt_raw, p.n);

  const uint32_t cellB = (uint32_t)C * (uint32_t)sizeof(FT);
  FT* orow = reinterpret_cast<FT*>(p.out.v[img]) + ((size_t)r * P + py) * P * C;
  const rsrc_t out = make_rsrc(orow, (uint32_t)P * cellB);
  const uint32_t vlane0 = (uint32_t)lane * Cell<FT>::LANE_BYTES;
  if (r >= cnt) {      // padded rows of the static-shape output are zero
    for (int px = 0; px < P; ++px)
      for (int c = lane * 4 + ch0, ch = ch0; ch < ch1; c += 256, ch += 256)
        if (c < C)
          Cell<FT>::store(out, vlane0 + (uint32_t)ch * (Cell<FT>::LANE_BYTES / 4), (uint32_t)px * cellB,
                          make_float4(0, 0, 0, 0));
    return;
  }

  const int lvl = rfl_i(min(max(lvl_raw, 0), p.num_levels - 1));
  const int H = p.H[lvl], W = p.W[lvl];
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

  // ---- the wave's sample rows (uniform) and, in lane px, the sample columns of bin column px
  const Tap ty0 = make_tap<PAD>(ay, py * S, crop, Hdim);
  const Tap ty1 = make_tap<PAD>(ay, py * S + (S - 1), crop, Hdim);
  const int bx = min(lane, P - 1);
  const Tap tx0 = make_tap<PAD>(ax, bx * S, crop, Wdim);
  const Tap tx1 = make_tap<PAD>(ax, bx * S + (S - 1), crop, Wdim);

  RowCtx rc;
  rc.feat = make_rsrc(p.data[img][lvl], (uint32_t)H * (uint32_t)W * cellB);
  const uint32_t rowB = (uint32_t)W * cellB;
  rc.yw[0] = rfl_f(ty0.lerp); rc.yw[1] = rfl_f(ty1.lerp);
  rc.vlane = vlane0;
  rc.cellB = cellB;
  const uint32_t yok = rfl_u((ty0.ok ? 1u : 0u) | (ty1.ok ? 2u : 0u));
  const uint32_t xok_l = (tx0.ok ? 1u : 0u) | (tx1.ok ? 2u : 0u);
  const uint32_t cabs_l[4] = {(uint32_t)tx0.lo * cellB, (uint32_t)tx0.hi * cellB, (uint32_t)tx1.lo * cellB,
                              (uint32_t)tx1.hi * cellB};
  const float xw0_l = tx0.lerp, xw1_l = tx1.lerp;
  // the four tapped rows as they are
  rc.row[0] = rfl_u((uint32_t)ty0.lo * rowB); rc.row[1] = rfl_u((uint32_t)ty0.hi * rowB);
  rc.row[2] = rfl_u((uint32_t)ty1.lo * rowB); rc.row[3] = rfl_u((uint32_t)ty1.hi * rowB);

  if (S == 1) {
    roi_row_single<FT>(rc, out, P, C, ch0, ch1, cabs_l, xok_l, xw0_l, yok);
    return;
  }
  // every sample of the row inside the map (boxes clipped to the image: nearly always)?
  const bool x_inside = __builtin_amdgcn_ballot_w64(lane < P && xok_l != 3) == 0;
  if (yok != 3 || !x_inside) {
    roi_row_guarded<POOL, FT>(rc, out, P, C, ch0, ch1, cabs_l, xok_l, xw0_l, xw1_l, yok);
    return;
  }
  const int ycls = rfl_i(PAD ? 2 : share_class(ty0, ty1));
  const uint32_t xcls_l = PAD ? 2u : (uint32_t)share_class(tx0, tx1);
  const uint32_t c0_l = cabs_l[0];
  const uint32_t crel_l[4] = {0u, cabs_l[1] - c0_l, cabs_l[2] - c0_l, cabs_l[3] - c0_l};
  if (ycls < 2) {      // sharing classes: row[i] = first row + i * row pitch
    rc.row[1] = rc.row[0] + rowB; rc.row[2] = rc.row[1] + rowB; rc.row[3] = rc.row[2];
  }
  // (float32 maps of a multiple of 256 channels; the float16 kernel is bound by its conversion arithmetic, the
  // carried registers cost it 20 %)
  if ((C & 255) == 0 && std::is_same<FT, float>::value) {
    if (ycls == 0) roi_row_carry<POOL, 0, FT>(rc, out, P, C, ch0, ch1, xcls_l, c0_l, crel_l, xw0_l, xw1_l);
    else if (ycls == 1) roi_row_carry<POOL, 1, FT>(rc, out, P, C, ch0, ch1, xcls_l, c0_l, crel_l, xw0_l, xw1_l);
    else roi_row_carry<POOL, 2, FT>(rc, out, P, C, ch0, ch1, xcls_l, c0_l, crel_l, xw0_l, xw1_l);
  } else if ((C & 255) == 0) {
    if (ycls == 0) roi_row<POOL, 0, true, FT>(rc, out, P, C, ch0, ch1, xcls_l, c0_l, crel_l, xw0_l, xw1_l);
    else if (ycls == 1) roi_row<POOL, 1, true, FT>(rc, out, P, C, ch0, ch1, xcls_l, c0_l, crel_l, xw0_l, xw1_l);
    else roi_row<POOL, 2, true, FT>(rc, out, P, C, ch0, ch1, xcls_l, c0_l, crel_l, xw0_l, xw1_l);
  } else {
    if (ycls == 0) roi_row<POOL, 0, false, FT>(rc, out, P, C, ch0, ch1, xcls_l, c0_l, crel_l, xw0_l, xw1_l);
    else if (ycls == 1) roi_row<POOL, 1, false, FT>(rc, out, P, C, ch0, ch1, xcls_l, c0_l, crel_l, xw0_l, xw1_l);
    else roi_row<POOL, 2, false, FT>(rc, out, P, C, ch0, ch1, xcls_l, c0_l, crel_l, xw0_l, xw1_l);
  }
}

// The float16 instantiations live in their own translation unit (roi_half.hip = this file with ODET_ROI_HALF_TU):
// roi.hip is compiled with -fno-slp-vectorize (hipcc's SLP pass packs the float32 lerps into v_pk_*_f32, which
// issue slower than the scalar forms), but the same flag costs the float16 kernel its packed conversions, so
// that one is built with SLP.
#define ODET_ROI_HALF_KERNELS(X)                                                                               \
  X(ODET_ROI_POOL_NONE, ODET_ROI_NORM_STRIDE) X(ODET_ROI_POOL_NONE, ODET_ROI_NORM_IMAGE)                       \
  X(ODET_ROI_POOL_NONE, ODET_ROI_NORM_TP_ALIGN) X(ODET_ROI_POOL_NONE, ODET_ROI_NORM_TP_ALIGN_NOPAD)            \
  X(ODET_ROI_POOL_MAX2, ODET_ROI_NORM_STRIDE) X(ODET_ROI_POOL_MAX2, ODET_ROI_NORM_IMAGE)                       \
  X(ODET_ROI_POOL_MAX2, ODET_ROI_NORM_TP_ALIGN) X(ODET_ROI_POOL_MAX2, ODET_ROI_NORM_TP_ALIGN_NOPAD)            \
  X(ODET_ROI_POOL_AVG2, ODET_ROI_NORM_STRIDE) X(ODET_ROI_POOL_AVG2, ODET_ROI_NORM_IMAGE)                       \
  X(ODET_ROI_POOL_AVG2, ODET_ROI_NORM_TP_ALIGN) X(ODET_ROI_POOL_AVG2, ODET_ROI_NORM_TP_ALIGN_NOPAD)
#ifdef ODET_ROI_HALF_TU
#define ODET_ROI_X(POOL, NORM)                                                       \
  template __global__ void k_roi_pool<POOL, NORM, __half, 0>(RoiParams);            \
  template __global__ void k_roi_pool<POOL, NORM, __half, 1>(RoiParams);
ODET_ROI_HALF_KERNELS(ODET_ROI_X)
#undef ODET_ROI_X
#else
#define ODET_ROI_X(POOL, NORM)                                                       \
  extern template __global__ void k_roi_pool<POOL, NORM, __half, 0>(RoiParams);     \
  extern template __global__ void k_roi_pool<POOL, NORM, __half, 1>(RoiParams);
ODET_ROI_HALF_KERNELS(ODET_ROI_X)
#undef ODET_ROI_X

template <int POOL, int NORM>
static void roi_launch(dim3 grid, int threads, hipStream_t st, const RoiParams& p, RoiEvents ev, int f16) {
  const bool timed = ev.start != nullptr || ev.stop != nullptr;
  if (f16) {
    if (timed)
      hipExtLaunchKernelGGL(HIP_KERNEL_NAME(k_roi_pool<POOL, NORM, __half, 1>), grid, dim3(threads), 0, st, ev.start,
                            ev.stop, 0, p);
    else
      hipLaunchKernelGGL(HIP_KERNEL_NAME(k_roi_pool<POOL, NORM, __half, 0>), grid, dim3(threads), 0, st, p);
  } else {
    if (timed)
      hipExtLaunchKernelGGL(HIP_KERNEL_NAME(k_roi_pool<POOL, NORM, float, 1>), grid, dim3(threads), 0, st, ev.start,
                            ev.stop, 0, p);
    else
      hipLaunchKernelGGL(HIP_KERNEL_NAME(k_roi_pool<POOL, NORM, float, 0>), grid, dim3(threads), 0, st, p);
  }
}

template <int POOL>
static void roi_launch_norm(int norm_mode, dim3 grid, int threads, hipStream_t st, const RoiParams& p, RoiEvents ev,
                            int f16) {
  switch (norm_mode) {
    case ODET_ROI_NORM_STRIDE: roi_launch<POOL, ODET_ROI_NORM_STRIDE>(grid, threads, st, p, ev, f16); break;
    case ODET_ROI_NORM_IMAGE: roi_launch<POOL, ODET_ROI_NORM_IMAGE>(grid, threads, st, p, ev, f16); break;
    case ODET_ROI_NORM_TP_ALIGN: roi_launch<POOL, ODET_ROI_NORM_TP_ALIGN>(grid, threads, st, p, ev, f16); break;
    default: roi_launch<POOL, ODET_ROI_NORM_TP_ALIGN_NOPAD>(grid, threads, st, p, ev, f16); break;
  }
}

// B images in one launch; all images share shapes and parameters
int odet_roi_pool_batch(const RoiImageIO* io, int B, int num_levels, int C, int n, int norm_mode, int image_h,
                        int image_w, int pool_size, int pool_mode, hipStream_t st, RoiEvents ev, int f16) {
  ODET_REQUIRE(n >= 0, "odet_roi_pool: negative n");
  if (n == 0) return ODET_OK;
  ODET_REQUIRE(io && B >= 1 && B <= ODET_MAX_BATCH, "odet_roi_pool: bad batch");
  ODET_REQUIRE(num_levels > 0 && num_levels <= ODET_MAX_LEVELS, "odet_roi_pool: num_levels %d out of range", num_levels);
  ODET_REQUIRE(C > 0 && (C & 3) == 0, "odet_roi_pool: C must be a positive multiple of 4 (got %d)", C);
  ODET_REQUIRE(norm_mode >= 0 && norm_mode <= 3, "odet_roi_pool: unknown norm_mode %d", norm_mode);
  ODET_REQUIRE(pool_mode >= 0 && pool_mode <= 2, "odet_roi_pool: unknown pool_mode %d", pool_mode);
  // a lane holds one bin column of the row
  ODET_REQUIRE(pool_size > 0 && pool_size <= 64, "odet_roi_pool: pool_size %d out of range (1..64)", pool_size);
  if (norm_mode == ODET_ROI_NORM_IMAGE) ODET_REQUIRE(image_h > 0 && image_w > 0, "odet_roi_pool: bad image shape");
  const size_t esz = f16 ? 2 : 4;
  RoiParams p;
  for (int i = 0; i < ODET_MAX_BATCH; ++i) {
    const RoiImageIO& a = io[i < B ? i : 0];
    ODET_REQUIRE(a.levels && a.rois && a.out, "odet_roi_pool: null pointer");
    ODET_REQUIRE(num_levels == 1 || a.roi_level, "odet_roi_pool: roi_level required with several levels");
    for (int l = 0; l < ODET_MAX_LEVELS; ++l) {
      const odet_level_t* L = &a.levels[l < num_levels ? l : 0];
      ODET_REQUIRE(L->data && L->H > 0 && L->W > 0, "odet_roi_pool: bad level %d", l);
      // cells are addressed with 32-bit byte offsets from the level's base
      ODET_REQUIRE((size_t)L->H * (size_t)L->W * (size_t)C * esz < (1ull << 31),
                   "odet_roi_pool: level %d is larger than 2 GiB", l);
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
  ODET_REQUIRE((size_t)pool_size * (size_t)C * esz < (1ull << 31), "odet_roi_pool: output row larger than 2 GiB");
  p.num_levels = num_levels;
  p.C = C; p.n = n; p.P = pool_size;
  p.image_h = (float)image_h; p.image_w = (float)image_w;
  // one wave per output row; a workgroup = the P rows of one RoI (8 rows of whatever RoIs when P > 16)
  p.waves = pool_size <= 16 ? pool_size : 8;
  const int64_t rows = (int64_t)n * pool_size;
  // a launch of a few hundred RoIs over 512 / 1024-channel maps (the C4 and VGG16 detectors: 300 RoIs) would leave
  // most CUs with one workgroup: split every RoI's channels over C / 256 workgroups
  p.slices = (p.waves == pool_size && (C & 255) == 0 && C > 256) ? C / 256 : 1;
  const int64_t blocks = p.slices > 1 ? (int64_t)n * p.slices : (rows + p.waves - 1) / p.waves;
  ODET_REQUIRE(blocks < (1ll << 30), "odet_roi_pool: too many workgroups");
  p.nblocks = (int)blocks;
  p.xcd_images = (B == 2 || B == 4 || B == 8) ? 1 : 0;
  p.xcds_per_img = p.xcd_images ? 8 / B : 8;
  p.blocks_per_xcd = (p.nblocks + p.xcds_per_img - 1) / p.xcds_per_img;   // per XCD of an image
  p.rois_per_xcd = 0;
  p.roi_groups = 0;
  if (p.slices > 1) {
    if (p.xcds_per_img % p.slices == 0) {                 // an XCD = one slice of one part of the RoIs
      p.roi_groups = p.xcds_per_img / p.slices;
      p.rois_per_xcd = (n + p.roi_groups - 1) / p.roi_groups;
      p.blocks_per_xcd = p.rois_per_xcd;
    } else {                                              // an XCD = all slices of its RoIs, slice-major
      p.rois_per_xcd = (n + p.xcds_per_img - 1) / p.xcds_per_img;
      p.blocks_per_xcd = p.rois_per_xcd * p.slices;
    }
    p.nblocks = p.blocks_per_xcd * p.xcds_per_img;        // (slots beyond the last RoI leave at `ri >= n`)
  }
  dim3 grid(p.blocks_per_xcd * 8, p.xcd_images ? 1 : B);
  const int threads = p.waves * 64;
  if (pool_mode == ODET_ROI_POOL_NONE) roi_launch_norm<ODET_ROI_POOL_NONE>(norm_mode, grid, threads, st, p, ev, f16);
  else if (pool_mode == ODET_ROI_POOL_MAX2) roi_launch_norm<ODET_ROI_POOL_MAX2>(norm_mode, grid, threads, st, p, ev, f16);
  else roi_launch_norm<ODET_ROI_POOL_AVG2>(norm_mode, grid, threads, st, p, ev, f16);
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
  auto make_key = [&](int r) -> unsigned long long { return d_roi_order_key(r, p.n, cnt, rois, lvl, p.inv_h, p.inv_w); };
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

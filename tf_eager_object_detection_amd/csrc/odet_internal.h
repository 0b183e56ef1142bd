// Internal helpers shared by the libodet_hip.so translation units (gfx950 only).
#ifndef ODET_INTERNAL_H_
#define ODET_INTERNAL_H_

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <mutex>

#include "../../include/odet.h"

#define ODET_WAVE 64

int odet_set_error(int code, const char* fmt, ...);

#define ODET_REQUIRE(cond, ...)                                   \
  do {                                                            \
    if (!(cond)) return odet_set_error(ODET_E_INVALID, __VA_ARGS__); \
  } while (0)

#define ODET_HIP(call)                                                                      \
  do {                                                                                      \
    hipError_t e_ = (call);                                                                 \
    if (e_ != hipSuccess)                                                                   \
      return odet_set_error(ODET_E_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                            __FILE__, __LINE__);                                            \
  } while (0)

#define ODET_LAUNCH_CHECK() ODET_HIP(hipGetLastError())

// Runs `fn() -> hipError_t` once per DEVICE of this process (keyed by the current device's ordinal) and returns its result every
// time: a kernel attribute (hipFuncSetAttribute: more than 64 KB of dynamic LDS) belongs to ONE device's copy of the kernel, so
// a process that drives a second GPU has to set it there too.  Thread-safe (executor threads may arrive together).
struct OdetPerDeviceOnce {
  static constexpr int MAXDEV = 64;
  std::once_flag once[MAXDEV];
  hipError_t rc[MAXDEV];
  template <typename F>
  hipError_t run(F fn) {
    int dev = 0;
    const hipError_t eg = hipGetDevice(&dev);
    if (eg != hipSuccess) return eg;
    if (dev < 0 || dev >= MAXDEV) return hipErrorInvalidDevice;
    std::call_once(once[dev], [&] { rc[dev] = fn(); });
    return rc[dev];
  }
};

static inline size_t odet_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Bump allocator over the caller's workspace.
struct OdetArena {
  char* base;
  size_t size;
  size_t off;
  template <typename T>
  T* take(size_t count) {
    size_t o = odet_align_up(off, 256);
    size_t end = o + count * sizeof(T);
    if (end > size) return nullptr;
    off = end;
    return reinterpret_cast<T*>(base + o);
  }
};

#ifdef __HIPCC__
// ---- device math -------------------------------------------------------------------------
// Correctly rounded float32 exp / log (through float64).  The oracle defines exp/log the
// same way, independently, with glibc.
__device__ __forceinline__ float d_exp32(float x) { return (float)exp((double)x); }
__device__ __forceinline__ float d_log32(float x) { return (float)log((double)x); }

// TF r1.13 non_max_suppression_op.cc IOUGreaterThanThreshold on corner-normalised boxes:
// b = (min0, min1, max0, max1) (element 0/2 and 1/3 already sorted).  No +1, strict '>',
// non-positive area never suppresses.  The quotient is a real float32 division.
__device__ __forceinline__ float4 d_norm_box(float4 b) {
  return make_float4(fminf(b.x, b.z), fminf(b.y, b.w), fmaxf(b.x, b.z), fmaxf(b.y, b.w));
}
__device__ __forceinline__ float d_box_area(float4 nb) { return (nb.z - nb.x) * (nb.w - nb.y); }
__device__ __forceinline__ bool d_iou_gt(float4 a, float area_a, float4 b, float area_b, float thr) {
  if (area_a <= 0.0f || area_b <= 0.0f) return false;
  float i0 = fmaxf(a.x, b.x), i1 = fmaxf(a.y, b.y);
  float i2 = fminf(a.z, b.z), i3 = fminf(a.w, b.w);
  float inter = fmaxf(i2 - i0, 0.0f) * fmaxf(i3 - i1, 0.0f);
  float iou = inter / (area_a + area_b - inter);
  return iou > thr;
}

// utils/bbox_transform.py:32-55 decode for one box (d = already  t*std+mean).
__device__ __forceinline__ float4 d_decode_box(float4 a, float d0, float d1, float d2, float d3) {
  float width = a.z - a.x + 1.0f;
  float height = a.w - a.y + 1.0f;
  float cx = a.x + 0.5f * width;
  float cy = a.y + 0.5f * height;
  cx = cx + d0 * width;
  cy = cy + d1 * height;
  width = width * d_exp32(d2);
  height = height * d_exp32(d3);
  float x1 = cx - 0.5f * width;
  float y1 = cy - 0.5f * height;
  return make_float4(x1, y1, x1 + width, y1 + height);
}

// utils/bbox_tf.py:70-74 clip.
__device__ __forceinline__ float4 d_clip_box(float4 b, float minv, float wmax, float hmax) {
  return make_float4(fmaxf(fminf(b.x, wmax), minv), fmaxf(fminf(b.y, hmax), minv),
                     fmaxf(fminf(b.z, wmax), minv), fmaxf(fminf(b.w, hmax), minv));
}

// order-preserving map float32 -> uint32 (ascending)
__device__ __forceinline__ uint32_t d_float_asc_key(float f) {
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
#endif  // __HIPCC__


// ---- shared parameter blocks / device helpers ------------------------------------------------
struct Vec4 { float v[4]; };

// Batched launches: up to ODET_MAX_BATCH independent images per kernel launch (blockIdx.y = image).
// Every image keeps its own caller-owned buffers, so a kernel receives a small table of pointers per
// buffer kind; all shapes / parameters are common to the batch.
#define ODET_MAX_BATCH 8
template <typename T>
struct PerImg {
  T v[ODET_MAX_BATCH];
};

// utils/anchor_generator.py:137-178 make_anchors over all pyramid levels (base_fpn_model.py:163-186).
struct FpnAnchorParams {
  int num_levels;
  int A;
  int fw[ODET_MAX_LEVELS];
  int stride[ODET_MAX_LEVELS];
  int start[ODET_MAX_LEVELS + 1];  // first anchor of each level
  float wh[ODET_MAX_LEVELS * ODET_MAX_ANCHORS_PER_CELL * 2];
};

#ifdef __HIPCC__
// anchor i of the concatenated level list: location-major / anchor-minor inside a level.
__device__ __forceinline__ float4 d_fpn_anchor(const FpnAnchorParams& p, int i) {
  int l = 0;
#pragma unroll
  for (int k = 1; k < ODET_MAX_LEVELS; ++k)
    if (k < p.num_levels && i >= p.start[k]) l = k;
  int j = i - p.start[l];
  int a = j % p.A;
  int cell = j / p.A;
  int x = cell % p.fw[l], y = cell / p.fw[l];
  float fs = (float)p.stride[l];
  float cx = (float)x * fs, cy = (float)y * fs;            // anchor_generator.py:146-147
  float hw = 0.5f * p.wh[(l * p.A + a) * 2 + 0];           // :160-161
  float hh = 0.5f * p.wh[(l * p.A + a) * 2 + 1];
  return make_float4(cx - hw, cy - hh, cx + hw, cy + hh);
}

// tf.nn.softmax arithmetic on a (bg, fg) pair: e = exp(x - max); p = e_fg * (1 / (e_bg + e_fg)).
__device__ __forceinline__ float d_fg_prob(float bg, float fg) {
  float m = fmaxf(bg, fg);
  float e0 = d_exp32(bg - m), e1 = d_exp32(fg - m);
  float inv = 1.0f / (e0 + e1);
  return e1 * inv;
}

// model/fpn/base_fpn_model.py:307-313 level of one RoI (0-based: level - min_level).
// The reference value is floor(4 + log(sqrt(w*h + 1e-8) / 224) / log(2)) in float32 with a correctly
// rounded log.  That log goes through float64 (~300 instructions); the hardware log2 gives the same
// quantity to ~1e-5, which decides the floor unless the value sits within 1e-4 of an integer -- only
// then is the exact expression evaluated, so the result is always the reference's.
__device__ __forceinline__ int d_roi_level(float4 b, int min_level, int max_level) {
  const float log2v = 0x1.62e43p-1f;   // (float)log(2.0): tf.log(2.) as a float32 constant
  float h = fmaxf(0.0f, b.w - b.y);                       // :307
  float w = fmaxf(0.0f, b.z - b.x);                       // :308
  const float x = sqrtf(w * h + 1e-8f) / 224.0f;
  float l;
  const float ya = 4.0f + __log2f(x);
  const float fl = floorf(ya);
  const float fr = ya - fl;
  if (fr > 1e-4f && fr < 1.0f - 1e-4f) {
    l = fl;
  } else {
    l = floorf(4.0f + d_log32(x) / log2v);                // :309
  }
  l = fmaxf(l, (float)min_level);                         // :312
  l = fminf(l, (float)max_level);                         // :313
  return (int)l - min_level;
}

__device__ __forceinline__ int wave_incl_scan(int v) {
  int lane = threadIdx.x & 63;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    int t = __shfl_up(v, d);
    if (lane >= d) v += t;
  }
  return v;
}

// exclusive scan across a block of up to 1024 threads; returns exclusive prefix, sets total.
__device__ __forceinline__ int block_excl_scan(int v, int* lds /*[17]*/, int* total) {
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int nw = (blockDim.x + 63) >> 6;
  int inc = wave_incl_scan(v);
  __syncthreads();
  if (lane == 63) lds[w] = inc;
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int k = 0; k < nw; ++k) { int t = lds[k]; lds[k] = run; run += t; }
    lds[16] = run;
  }
  __syncthreads();
  *total = lds[16];
  return lds[w] + inc - v;
}

// ---- in-workgroup sorts of 64-bit keys ---------------------------------------------------------
__device__ __forceinline__ void bitonic_sort_u64(unsigned long long* keys, int P2, int nthreads) {
  for (int k = 2; k <= P2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = threadIdx.x; t < (P2 >> 1); t += nthreads) {
        // index of the lower element of the t-th compare-exchange pair at distance j
        int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        int l = i | j;
        bool up = ((i & k) == 0);
        unsigned long long a = keys[i], b = keys[l];
        if ((a > b) == up) { keys[i] = b; keys[l] = a; }
      }
      __syncthreads();
    }
  }
}

// Two float32 -> one dword of two float16 (round to nearest even) in ONE instruction: gfx950's v_cvt_pk_f16_f32.  hipcc emits
// two v_cvt_f16_f32 and a v_pack_b32_f16 for the same thing; bit-identical over all 2^32 inputs (tools/exp/cvt_pk_probe.hip).
__device__ __forceinline__ unsigned d_cvt_pk_f16(float a, float b) {
  unsigned r;
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// eight float32 -> eight float16 (any 16-byte vector type)
template <typename H8>
__device__ __forceinline__ H8 d_cvt8_f16(const float* f) {
  typedef unsigned cvt_u4 __attribute__((ext_vector_type(4)));
  cvt_u4 r;
  r[0] = d_cvt_pk_f16(f[0], f[1]); r[1] = d_cvt_pk_f16(f[2], f[3]);
  r[2] = d_cvt_pk_f16(f[4], f[5]); r[3] = d_cvt_pk_f16(f[6], f[7]);
  return __builtin_bit_cast(H8, r);
}
// four such dwords as one 16-byte vector
template <typename H8>
__device__ __forceinline__ H8 d_pack8_f16(const unsigned* u) {
  typedef unsigned cvt_u4 __attribute__((ext_vector_type(4)));
  const cvt_u4 r = {u[0], u[1], u[2], u[3]};
  return __builtin_bit_cast(H8, r);
}

__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int mask) {
  uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, mask);
  uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), mask);
  return ((unsigned long long)hi << 32) | lo;
}

// Ascending bitonic sort of 1024 keys, one per thread of a 1024-thread workgroup; xch: LDS [2][1024].
__device__ __forceinline__ unsigned long long bitonic_sort_1024_reg(unsigned long long key, unsigned long long* xch) {
  const int tid = threadIdx.x;
  int buf = 0;
  for (int k = 2; k <= 1024; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      unsigned long long other;
      if (j >= 64) {
        unsigned long long* x = xch + buf * 1024;
        x[tid] = key;
        __syncthreads();
        other = x[tid ^ j];
        buf ^= 1;               // the next LDS stage writes the other buffer: one barrier per stage
      } else {
        other = shfl_xor_u64(key, j);
      }
      const bool up = ((tid & k) == 0);
      const bool lower = ((tid & j) == 0);
      const unsigned long long mn = key < other ? key : other, mx = key < other ? other : key;
      key = (lower == up) ? mn : mx;
    }
  }
  return key;
}


// Processing-order key of RoI row r for the RoI kernel (odet_roi_order): (level, y centre, x centre) quantised to
// 12 bits each, row index in the low word; rows >= cnt (padding of the static shape) sort last, rows >= n never.
__device__ __forceinline__ unsigned long long d_roi_order_key(int r, int n, int cnt, const float4* rois, const int32_t* lvl,
                                                               float inv_h, float inv_w) {
  if (r >= n) return ~0ull;
  if (r >= cnt) return (0xFFFFFFFEull << 32) | (unsigned)r;          // padded rows last (they are zero-filled)
  const float4 b = rois[r];
  const int l = lvl ? min(max(lvl[r], 0), 7) : 0;
  const int qy = min(max((int)((b.y + b.w) * 0.5f * inv_h * 4096.0f), 0), 4095);
  const int qx = min(max((int)((b.x + b.z) * 0.5f * inv_w * 4096.0f), 0), 4095);
  return ((unsigned long long)((l << 24) | (qy << 12) | qx) << 32) | (unsigned)r;
}

// model/fpn/base_fpn_model.py:303-324 _assign_levels by ONE workgroup of THREADS threads: level per RoI,
// then a stable partition by level (ascending original index inside a level) -- the order
// tf.where + tf.gather + tf.concat produce at :316-324.  rois may have been written by this
// workgroup (caller syncs first).  Thread t owns the ceil(cnt/THREADS) consecutive RoIs starting at
// t*per, so (wave, lane, item) order is index order; per level: wave-scan of the per-thread counts,
// per-wave totals through LDS, one small scan over (level, wave) by wave 0.  Two barriers.
template <int THREADS>
__device__ __forceinline__ void d_assign_levels_block(const float4* rois, int cnt, int min_level, int max_level,
                                                      float4* __restrict__ out_rois, int32_t* __restrict__ out_level,
                                                      int64_t* __restrict__ out_perm, int32_t* __restrict__ out_counts,
                                                      int* lds /*[ODET_MAX_LEVELS * 16 + ODET_MAX_LEVELS]*/) {
  constexpr int ITEMS = ODET_ASSIGN_MAX_ROIS / THREADS;
  constexpr int NW = THREADS / 64;
  const int nl = max_level - min_level + 1;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int per = (cnt + THREADS - 1) / THREADS;
  const int lo = threadIdx.x * per;
  int lv[ITEMS];
  float4 rb[ITEMS];
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    const int i = lo + k;
    lv[k] = -1;
    if (k < per && i < cnt) { rb[k] = rois[i]; lv[k] = d_roi_level(rb[k], min_level, max_level); }
  }
  int off[ODET_MAX_LEVELS];   // exclusive offset of this thread inside its wave, per level
#pragma unroll
  for (int L = 0; L < ODET_MAX_LEVELS; ++L) {
    off[L] = 0;
    if (L < nl) {
      int c = 0;
#pragma unroll
      for (int k = 0; k < ITEMS; ++k) c += (lv[k] == L) ? 1 : 0;
      const int inc = wave_incl_scan(c);
      off[L] = inc - c;
      if (lane == 63) lds[L * 16 + w] = inc;
    }
  }
  __syncthreads();
  if (w == 0) {
    int base = 0;
    for (int L = 0; L < nl; ++L) {
      const int v = (lane < NW) ? lds[L * 16 + lane] : 0;
      const int inc = wave_incl_scan(v);
      const int total = __shfl(inc, NW - 1);
      if (lane < NW) lds[L * 16 + lane] = base + inc - v;
      if (lane == 0) { lds[ODET_MAX_LEVELS * 16 + L] = total; out_counts[L] = total; }
      base += total;
    }
  }
  __syncthreads();
#pragma unroll
  for (int L = 0; L < ODET_MAX_LEVELS; ++L) {
    if (L < nl) {
      int pos = lds[L * 16 + w] + off[L];
#pragma unroll
      for (int k = 0; k < ITEMS; ++k) {
        if (lv[k] == L) {
          out_rois[pos] = rb[k];
          out_level[pos] = L;
          out_perm[pos] = lo + k;
          ++pos;
        }
      }
    }
  }
}
#endif  // __HIPCC__

struct Float4Host { float v[4]; };

// ---- batched internals shared between translation units (executor.hip drives them) -----------
struct FpnProposalIO {      // per-image arguments of the FPN proposal stage
  const float* rpn_logits; const float* rpn_deltas;
  float* out_rois; int32_t* out_idx; int32_t* out_count;
  float* out_sorted_rois; int32_t* out_level; int64_t* out_perm; int32_t* out_level_counts;
  int32_t* out_done; void* workspace; size_t workspace_bytes;
  int32_t* out_order;       // nullable: spatial processing order of the level-sorted RoIs (odet_roi_order), written by
                            // the NMS walk's tail when max_output <= 1024
};
int odet_fpn_proposals_batch(const FpnProposalIO* io, int B, int num_levels, int A, const int* fh, const int* fw,
                             const int* stride, const float* wh, int image_h, int image_w, const float* means,
                             const float* stds, int max_output, float iou_threshold, int min_level, int max_level,
                             int blind_chunks, hipStream_t st, int first_chunk = 0, int ws_clean = 0);

int odet_frcnn_proposals_batch(const FpnProposalIO* io, int B, const float* anchor_base, int A, int feat_stride, int fh,
                               int fw, int image_h, int image_w, const float* means, const float* stds, int max_output,
                               float iou_threshold, int blind_chunks, hipStream_t st, int first_chunk = 0, int ws_clean = 0);

struct RoiEvents { hipEvent_t start, stop; };   // optional: timestamps of the dispatch itself
struct RoiImageIO {         // per-image arguments (order: nullable processing order, odet_roi_order)
  const odet_level_t* levels; const float* rois; const int32_t* roi_level; const int32_t* count_dev;
  const int32_t* order; float* out;
};
struct RoiOrderIO { const float* rois; const int32_t* roi_level; const int32_t* count_dev; int32_t* order; };
int odet_roi_order_batch(const RoiOrderIO* io, int B, int n, int image_h, int image_w, hipStream_t st);
int odet_roi_pool_batch(const RoiImageIO* io, int B, int num_levels, int C, int n, int norm_mode, int image_h,
                        int image_w, int pool_size, int pool_mode, hipStream_t st, RoiEvents ev, int f16 = 0);

struct PostOpsExtra { float wmax, hmax, roi_div; int mode; };
struct PostOpsImageIO {     // per-image arguments
  const float* scores; const float* deltas; const float* rois; const int32_t* count_dev;
  float* out_boxes; int32_t* out_labels; float* out_scores; int32_t* out_count; float* out_record;
  void* workspace; size_t workspace_bytes;
};
int odet_post_ops_batch(const PostOpsImageIO* io, int B, int R, int Ccls, int num_classes, PostOpsExtra ex,
                        const float* means, const float* stds, int max_per_class, int max_per_image,
                        float nms_iou_threshold, float score_threshold, float min_edge, hipStream_t st, int ws_clean);

// shared between translation units
struct OdetSortImage {     // one image of odet_sort_keys_desc_batch
  uint32_t* keys_a; uint32_t* vals_a; uint32_t* keys_b; uint32_t* vals_b; uint32_t* hist; const int32_t* skip;
};
int odet_sort_keys_desc_batch(int n, int B, const OdetSortImage* imgs, hipStream_t st);
int odet_sort_keys_desc(int n, uint32_t* keys_a, uint32_t* vals_a, uint32_t* keys_b, uint32_t* vals_b,
                        uint32_t* hist, const int32_t* skip, uint32_t** sorted_vals, hipStream_t stream);
size_t odet_sort_hist_entries(int n);

#endif  // ODET_INTERNAL_H_

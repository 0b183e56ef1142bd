// Internal helpers shared by the libodet_hip.so translation units (gfx950 only).
#ifndef ODET_INTERNAL_H_
#define ODET_INTERNAL_H_

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

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

struct Float4Host { float v[4]; };

// shared between translation units
int odet_sort_pairs_desc(const float* scores, int n, uint32_t* keys_a, uint32_t* vals_a,
                         uint32_t* keys_b, uint32_t* vals_b, uint32_t* hist, int32_t* n_invalid_dev,
                         uint32_t** sorted_vals, hipStream_t stream);
size_t odet_sort_hist_entries(int n);

#endif  // ODET_INTERNAL_H_

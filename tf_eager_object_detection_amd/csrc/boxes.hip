// Anchors, box transforms, filters, IoU, gather, RPN score glue, FPN level assignment.
// All kernels here are HBM/launch-bound elementwise or compaction work: one box (16 B) per
// lane, float4 loads/stores so a wave moves 1 KiB per instruction.
#include <stdarg.h>

#include "odet_internal.h"

// ------------------------------------------------------------------------------ errors --
static thread_local char g_err[512] = "";

int odet_set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

extern "C" int odet_version(void) { return ODET_VERSION; }
extern "C" const char* odet_last_error(void) { return g_err; }

static inline int grid_for(int64_t n, int block) { return (int)((n + block - 1) / block); }

// ----------------------------------------------------------------------------- anchors --
__global__ void __launch_bounds__(256) k_anchors_shift(const float4* __restrict__ base, int A, int stride,
                                                       int fw, int total, float4* __restrict__ out) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  int a = i % A;
  int cell = i / A;
  int x = cell % fw, y = cell / fw;
  float sx = (float)(x * stride), sy = (float)(y * stride);
  float4 b = base[a];
  out[i] = make_float4(b.x + sx, b.y + sy, b.z + sx, b.w + sy);
}

extern "C" int odet_anchors_shift(const float* anchor_base, int A, int feat_stride, int fh, int fw,
                                  float* out, odet_stream_t stream) {
  ODET_REQUIRE(anchor_base && out, "odet_anchors_shift: null pointer");
  ODET_REQUIRE(A > 0 && fh >= 0 && fw >= 0 && feat_stride > 0, "odet_anchors_shift: bad sizes");
  int64_t total = (int64_t)fh * fw * A;
  ODET_REQUIRE(total < (1ll << 31), "odet_anchors_shift: too many anchors");
  if (total == 0) return ODET_OK;
  hipLaunchKernelGGL(k_anchors_shift, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const float4*)anchor_base, A, feat_stride, fw, (int)total, (float4*)out);
  ODET_LAUNCH_CHECK();
  return ODET_OK;
}

__global__ void __launch_bounds__(256) k_anchors_fpn(FpnAnchorParams p, float4* __restrict__ out) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= p.start[p.num_levels]) return;
  out[i] = d_fpn_anchor(p, i);
}

extern "C" int odet_anchors_fpn(int num_levels, int A, const int* fh, const int* fw, const int* stride,
                                const float* wh, float* out, odet_stream_t stream) {
  ODET_REQUIRE(fh && fw && stride && wh && out, "odet_anchors_fpn: null pointer");
  ODET_REQUIRE(num_levels > 0 && num_levels <= ODET_MAX_LEVELS, "odet_anchors_fpn: num_levels %d out of range", num_levels);
  ODET_REQUIRE(A > 0 && A <= ODET_MAX_ANCHORS_PER_CELL, "odet_anchors_fpn: A %d out of range", A);
  FpnAnchorParams p;
  p.num_levels = num_levels;
  p.A = A;
  int64_t total = 0;
  for (int l = 0; l < num_levels; ++l) {
    ODET_REQUIRE(fh[l] >= 0 && fw[l] > 0 && stride[l] > 0, "odet_anchors_fpn: bad level %d", l);
    p.fw[l] = fw[l];
    p.stride[l] = stride[l];
    p.start[l] = (int)total;
    total += (int64_t)fh[l] * fw[l] * A;
    ODET_REQUIRE(total < (1ll << 31), "odet_anchors_fpn: too many anchors");
  }
  for (int l = num_levels; l <= ODET_MAX_LEVELS; ++l) p.start[l] = (int)total;
  p.start[num_levels] = (int)total;
  for (int i = 0; i < num_levels * A * 2; ++i) p.wh[i] = wh[i];
  if (total == 0) return ODET_OK;
  hipLaunchKernelGGL(k_anchors_fpn, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, p,
                     (float4*)out);
  ODET_LAUNCH_CHECK();
  return ODET_OK;
}

// ---------------------------------------------------------------------- decode / encode --

__global__ void __launch_bounds__(256) k_decode(const float4* __restrict__ anchors, const float* __restrict__ deltas,
                                                int64_t delta_stride, int n, Vec4 means, Vec4 stds, int clip,
                                                float wmax, float hmax, float4* __restrict__ out) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float* t = deltas + (int64_t)i * delta_stride;
  float4 d;
  if ((delta_stride & 3) == 0) {
    d = *reinterpret_cast<const float4*>(t);
  } else {
    d = make_float4(t[0], t[1], t[2], t[3]);
  }
  float d0 = d.x * stds.v[0] + means.v[0];   // bbox_transform.py:37
  float d1 = d.y * stds.v[1] + means.v[1];
  float d2 = d.z * stds.v[2] + means.v[2];
  float d3 = d.w * stds.v[3] + means.v[3];
  float4 b = d_decode_box(anchors[i], d0, d1, d2, d3);
  if (clip) b = d_clip_box(b, 0.0f, wmax, hmax);
  out[i] = b;
}

extern "C" int odet_decode(const float* anchors, const float* deltas, int64_t delta_stride, int n,
                           const float* means, const float* stds, int clip_h, int clip_w, float* out,
                           odet_stream_t stream) {
  ODET_REQUIRE(n >= 0, "odet_decode: negative n");
  if (n == 0) return ODET_OK;
  ODET_REQUIRE(anchors && deltas && means && stds && out, "odet_decode: null pointer");
  ODET_REQUIRE(delta_stride >= 4, "odet_decode: delta_stride must be >= 4");
  Vec4 m, s;
  for (int k = 0; k < 4; ++k) { m.v[k] = means[k]; s.v[k] = stds[k]; }
  hipLaunchKernelGGL(k_decode, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const float4*)anchors, deltas, delta_stride, n, m, s, clip_h > 0 ? 1 : 0,
                     (float)(clip_w - 1), (float)(clip_h - 1), (float4*)out);
  ODET_LAUNCH_CHECK();
  return ODET_OK;
}

__global__ void __launch_bounds__(256) k_encode(const float4* __restrict__ src, const float4* __restrict__ dst, int n,
                                                Vec4 means, Vec4 stds, float4* __restrict__ out) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float4 b = src[i], g = dst[i];
  float width = b.z - b.x + 1.0f, height = b.w - b.y + 1.0f;     // bbox_transform.py:11-14
  float cx = b.x + 0.5f * width, cy = b.y + 0.5f * height;
  float gw = g.z - g.x + 1.0f, gh = g.w - g.y + 1.0f;            // :16-19
  float gcx = g.x + 0.5f * gw, gcy = g.y + 0.5f * gh;
  float dx = (gcx - cx) / width;                                 // :21-24
  float dy = (gcy - cy) / height;
  float dw = d_log32(gw / width);
  float dh = d_log32(gh / height);
  out[i] = make_float4((dx - means.v[0]) / stds.v[0], (dy - means.v[1]) / stds.v[1],
                       (dw - means.v[2]) / stds.v[2], (dh - means.v[3]) / stds.v[3]);   // :27
}

extern "C" int odet_encode(const float* src, const float* dst, int n, const float* means, const float* stds,
                           float* out, odet_stream_t stream) {
  ODET_REQUIRE(n >= 0, "odet_encode: negative n");
  if (n == 0) return ODET_OK;
  ODET_REQUIRE(src && dst && means && stds && out, "odet_encode: null pointer");
  Vec4 m, s;
  for (int k = 0; k < 4; ++k) { m.v[k] = means[k]; s.v[k] = stds[k]; }
  hipLaunchKernelGGL(k_encode, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const float4*)src, (const float4*)dst, n, m, s, (float4*)out);
  ODET_LAUNCH_CHECK();
  return ODET_OK;
}

__global__ void __launch_bounds__(256) k_clip(const float4* __restrict__ in, int n, float minv, float wmax, float hmax,
                                              float4* __restrict__ out) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  out[i] = d_clip_box(in[i], minv, wmax, hmax);
}

extern "C" int odet_clip(const float* boxes, int n, float min_value, int max_h, int max_w, float* out,
                         odet_stream_t stream) {
  ODET_REQUIRE(n >= 0, "odet_clip: negative n");
  if (n == 0) return ODET_OK;
  ODET_REQUIRE(boxes && out, "odet_clip: null pointer");
  hipLaunchKernelGGL(k_clip, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, (const float4*)boxes, n,
                     min_value, (float)(max_w - 1), (float)(max_h - 1), (float4*)out);
  ODET_LAUNCH_CHECK();
  return ODET_OK;
}

// -------------------------------------------------------------------- ordered compaction --
// Three launches: (1) per-element predicate -> flag byte + per-block count, (2) one-workgroup
// exclusive scan of the block counts, (3) ordered write.  Ascending-index order is part of the
// contract (tf.where semantics).
#define CP_BLOCK 256
#define CP_ITEMS 4
#define CP_TILE (CP_BLOCK * CP_ITEMS)

enum { CP_CLIP_FILTER = 0, CP_RANGE = 1, CP_WHERE_GT = 2 };

struct CompactArgs {
  const float* data;   // boxes [n,4] or strided scalar values
  int64_t stride;      // WHERE_GT: element stride
  int n;
  float minv, wmax, hmax, min_edge, thr;
};

template <int MODE>
__device__ __forceinline__ bool cp_pred(const CompactArgs& a, int i) {
  if (MODE == CP_WHERE_GT) {
    return a.data[(int64_t)i * a.stride] > a.thr;
  } else {
    float4 b = reinterpret_cast<const float4*>(a.data)[i];
    if (MODE == CP_RANGE) {
      return b.x >= 0.0f && b.y >= 0.0f && b.z <= a.wmax && b.w <= a.hmax;   // bbox_tf.py:94-99
    } else {
      float4 c = d_clip_box(b, a.minv, a.wmax, a.hmax);
      float e0 = c.z - c.x + 1.0f;   // bbox_tf.py:81-82 (names swapped there, test symmetric)
      float e1 = c.w - c.y + 1.0f;
      return e1 >= a.min_edge && e0 >= a.min_edge;
    }
  }
}

template <int MODE>
__global__ void __launch_bounds__(CP_BLOCK) k_cp_flags(CompactArgs a, uint8_t* __restrict__ flags,
                                                       int* __restrict__ blockcnt) {
  __shared__ int lds[17];
  int base = blockIdx.x * CP_TILE + threadIdx.x * CP_ITEMS;
  int c = 0;
#pragma unroll
  for (int k = 0; k < CP_ITEMS; ++k) {
    int i = base + k;
    if (i < a.n) {
      bool f = cp_pred<MODE>(a, i);
      flags[i] = f ? 1 : 0;
      c += f ? 1 : 0;
    }
  }
  int total;
  block_excl_scan(c, lds, &total);
  if (threadIdx.x == 0) blockcnt[blockIdx.x] = total;
}

__global__ void __launch_bounds__(1024) k_cp_scan(int* __restrict__ blockcnt, int nblocks, int32_t* __restrict__ out_count) {
  __shared__ int lds[17];
  int per = (nblocks + 1023) / 1024;
  int lo = threadIdx.x * per;
  int s = 0;
  for (int k = 0; k < per; ++k)
    if (lo + k < nblocks) s += blockcnt[lo + k];
  int total;
  int ex = block_excl_scan(s, lds, &total);
  for (int k = 0; k < per; ++k)
    if (lo + k < nblocks) { int t = blockcnt[lo + k]; blockcnt[lo + k] = ex; ex += t; }
  if (threadIdx.x == 0) *out_count = total;
}

template <int MODE>
__global__ void __launch_bounds__(CP_BLOCK) k_cp_write(CompactArgs a, const uint8_t* __restrict__ flags,
                                                       const int* __restrict__ blockoff, int64_t* __restrict__ out_idx,
                                                       float4* __restrict__ out_boxes) {
  __shared__ int lds[17];
  int base = blockIdx.x * CP_TILE + threadIdx.x * CP_ITEMS;
  uint8_t f[CP_ITEMS];
  int c = 0;
#pragma unroll
  for (int k = 0; k < CP_ITEMS; ++k) {
    int i = base + k;
    f[k] = (i < a.n) ? flags[i] : 0;
    c += f[k];
  }
  int total;
  int pos = blockoff[blockIdx.x] + block_excl_scan(c, lds, &total);
#pragma unroll
  for (int k = 0; k < CP_ITEMS; ++k) {
    if (f[k]) {
      int i = base + k;
      out_idx[pos] = i;
      if (MODE == CP_CLIP_FILTER)
        out_boxes[pos] = d_clip_box(reinterpret_cast<const float4*>(a.data)[i], a.minv, a.wmax, a.hmax);
      ++pos;
    }
  }
}

extern "C" size_t odet_compact_workspace_bytes(int n) {
  if (n < 0) n = 0;
  size_t nblocks = ((size_t)n + CP_TILE - 1) / CP_TILE;
  return odet_align_up((size_t)n, 256) + odet_align_up((nblocks + 1) * sizeof(int), 256) + 512;
}

template <int MODE>
static int run_compact(const CompactArgs& a, int64_t* out_idx, float4* out_boxes, int32_t* out_count, void* ws,
                       size_t ws_bytes, hipStream_t st, const char* who) {
  ODET_REQUIRE(a.n >= 0, "%s: negative n", who);
  ODET_REQUIRE(out_idx && out_count, "%s: null output", who);
  if (a.n == 0) {
    ODET_HIP(hipMemsetAsync(out_count, 0, sizeof(int32_t), st));
    return ODET_OK;
  }
  ODET_REQUIRE(a.data, "%s: null input", who);
  if (!ws || ws_bytes < odet_compact_workspace_bytes(a.n))
    return odet_set_error(ODET_E_WORKSPACE, "%s: workspace too small (%zu < %zu)", who, ws_bytes,
                          odet_compact_workspace_bytes(a.n));
  OdetArena ar{(char*)ws, ws_bytes, 0};
  int nblocks = (a.n + CP_TILE - 1) / CP_TILE;
  uint8_t* flags = ar.take<uint8_t>(a.n);
  int* blockcnt = ar.take<int>(nblocks + 1);
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cp_flags<MODE>), dim3(nblocks), dim3(CP_BLOCK), 0, st, a, flags, blockcnt);
  ODET_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_cp_scan, dim3(1), dim3(1024), 0, st, blockcnt, nblocks, out_count);
  ODET_LAUNCH_CHECK();
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cp_write<MODE>), dim3(nblocks), dim3(CP_BLOCK), 0, st, a, flags, blockcnt,
                     out_idx, out_boxes);
  ODET_LAUNCH_CHECK();
  return ODET_OK;
}

extern "C" int odet_clip_filter(const float* boxes, int n, float min_value, int max_h, int max_w, float min_edge,
                                float* out_boxes, int64_t* out_idx, int32_t* out_count, void* workspace,
                                size_t workspace_bytes, odet_stream_t stream) {
  ODET_REQUIRE(out_boxes, "odet_clip_filter: null out_boxes");
  CompactArgs a{boxes, 4, n, min_value, (float)(max_w - 1), (float)(max_h - 1), min_edge, 0.0f};
  return run_compact<CP_CLIP_FILTER>(a, out_idx, (float4*)out_boxes, out_count, workspace, workspace_bytes,
                                     (hipStream_t)stream, "odet_clip_filter");
}

extern "C" int odet_range_filter(const float* boxes, int n, int max_h, int max_w, int64_t* out_idx,
                                 int32_t* out_count, void* workspace, size_t workspace_bytes, odet_stream_t stream) {
  CompactArgs a{boxes, 4, n, 0.0f, (float)(max_w - 1), (float)(max_h - 1), 0.0f, 0.0f};
  return run_compact<CP_RANGE>(a, out_idx, nullptr, out_count, workspace, workspace_bytes, (hipStream_t)stream,
                               "odet_range_filter");
}

extern "C" int odet_where_greater(const float* values, int64_t stride, int n, float thr, int64_t* out_idx,
                                  int32_t* out_count, void* workspace, size_t workspace_bytes, odet_stream_t stream) {
  ODET_REQUIRE(stride >= 1, "odet_where_greater: stride must be >= 1");
  CompactArgs a{values, stride, n, 0.0f, 0.0f, 0.0f, 0.0f, thr};
  return run_compact<CP_WHERE_GT>(a, out_idx, nullptr, out_count, workspace, workspace_bytes, (hipStream_t)stream,
                                  "odet_where_greater");
}

// ------------------------------------------------------------------------- pairwise IoU --
// One wave covers 64 consecutive columns j of one row i: boxes2 reads and the output row are
// coalesced; boxes1[i] is a wave-uniform (scalar) load.
__global__ void __launch_bounds__(256) k_pairwise_iou(const float4* __restrict__ b1, int n, const float4* __restrict__ b2,
                                                      int m, float* __restrict__ out) {
  int j = blockIdx.y * 64 + (threadIdx.x & 63);
  int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n || j >= m) return;
  float4 p = b1[i], q = b2[j];
  float a1 = (p.w - p.y + 1.0f) * (p.z - p.x + 1.0f);            // bbox_tf.py:14-15
  float a2 = (q.w - q.y + 1.0f) * (q.z - q.x + 1.0f);
  float ih = fmaxf(0.0f, fminf(p.w, q.w) - fmaxf(p.y, q.y) + 1.0f);   // :28-30
  float iw = fmaxf(0.0f, fminf(p.z, q.z) - fmaxf(p.x, q.x) + 1.0f);   // :31-33
  float inter = ih * iw;
  float uni = a1 + a2 - inter;                                   // :51-52
  out[(int64_t)i * m + j] = (inter == 0.0f) ? 0.0f : inter / uni;   // :54-56
}

extern "C" int odet_pairwise_iou(const float* boxes1, int n, const float* boxes2, int m, float* out,
                                 odet_stream_t stream) {
  ODET_REQUIRE(n >= 0 && m >= 0, "odet_pairwise_iou: negative size");
  if (n == 0 || m == 0) return ODET_OK;
  ODET_REQUIRE(boxes1 && boxes2 && out, "odet_pairwise_iou: null pointer");
  ODET_REQUIRE((m + 63) / 64 <= 65535, "odet_pairwise_iou: m too large for one launch (max 4194240)");
  hipLaunchKernelGGL(k_pairwise_iou, dim3((n + 3) / 4, (m + 63) / 64), dim3(256), 0, (hipStream_t)stream,
                     (const float4*)boxes1, n, (const float4*)boxes2, m, out);
  ODET_LAUNCH_CHECK();
  return ODET_OK;
}

// ------------------------------------------------------------------------------- gather --
template <typename IdxT>
__global__ void __launch_bounds__(256) k_gather_rows(const float* __restrict__ src, const IdxT* __restrict__ idx, int n,
                                                     const int32_t* __restrict__ count_dev, int row_floats,
                                                     float* __restrict__ out) {
  int cnt = count_dev ? min(*count_dev, n) : n;
  int64_t total = (int64_t)cnt * row_floats;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    int r = (int)(e / row_floats);
    int c = (int)(e % row_floats);
    out[e] = src[(int64_t)idx[r] * row_floats + c];
  }
}

extern "C" int odet_gather_rows(const float* src, const void* idx, int idx_is_64, int n, const int32_t* count_dev,
                                int row_floats, float* out, odet_stream_t stream) {
  ODET_REQUIRE(n >= 0 && row_floats > 0, "odet_gather_rows: bad sizes");
  if (n == 0) return ODET_OK;
  ODET_REQUIRE(src && idx && out, "odet_gather_rows: null pointer");
  int64_t total = (int64_t)n * row_floats;
  int grid = (int)((total + 255) / 256);
  if (grid > 4096) grid = 4096;
  if (idx_is_64)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gather_rows<int64_t>), dim3(grid), dim3(256), 0, (hipStream_t)stream, src,
                       (const int64_t*)idx, n, count_dev, row_floats, out);
  else
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gather_rows<int32_t>), dim3(grid), dim3(256), 0, (hipStream_t)stream, src,
                       (const int32_t*)idx, n, count_dev, row_floats, out);
  ODET_LAUNCH_CHECK();
  return ODET_OK;
}

// ----------------------------------------------------------------------- RPN fg softmax --
__global__ void __launch_bounds__(256) k_rpn_fg_fpn(const float2* __restrict__ logits, int n, float* __restrict__ out) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float2 l = logits[i];
  out[i] = d_fg_prob(l.x, l.y);
}

__global__ void __launch_bounds__(256) k_rpn_fg_frcnn(const float* __restrict__ logits, int nloc, int A,
                                                      float* __restrict__ out) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= nloc * A) return;
  int l = i / A, a = i % A;
  const float* row = logits + (int64_t)l * 2 * A;
  out[i] = d_fg_prob(row[a], row[A + a]);
}

extern "C" int odet_rpn_fg_softmax(const float* logits, int nloc, int A, int layout, float* out,
                                   odet_stream_t stream) {
  ODET_REQUIRE(nloc >= 0, "odet_rpn_fg_softmax: negative size");
  if (nloc == 0) return ODET_OK;
  ODET_REQUIRE(logits && out, "odet_rpn_fg_softmax: null pointer");
  if (layout == ODET_RPN_LAYOUT_FPN) {
    hipLaunchKernelGGL(k_rpn_fg_fpn, dim3(grid_for(nloc, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float2*)logits, nloc, out);
  } else if (layout == ODET_RPN_LAYOUT_FRCNN) {
    ODET_REQUIRE(A > 0 && (int64_t)nloc * A < (1ll << 31), "odet_rpn_fg_softmax: bad A");
    hipLaunchKernelGGL(k_rpn_fg_frcnn, dim3(grid_for((int64_t)nloc * A, 256)), dim3(256), 0, (hipStream_t)stream,
                       logits, nloc, A, out);
  } else {
    return odet_set_error(ODET_E_INVALID, "odet_rpn_fg_softmax: unknown layout %d", layout);
  }
  ODET_LAUNCH_CHECK();
  return ODET_OK;
}

// ----------------------------------------------------------------------- assign levels --
// One workgroup: level per RoI, then a stable partition by level (ascending original index
// inside a level) -- the order tf.where + tf.gather + tf.concat produce in
// base_fpn_model.py:316-324.
#define AL_THREADS 1024

__global__ void __launch_bounds__(AL_THREADS) k_assign_levels(const float4* __restrict__ rois, int n,
                                                              const int32_t* __restrict__ count_dev, int min_level,
                                                              int max_level, float4* __restrict__ out_rois,
                                                              int32_t* __restrict__ out_level,
                                                              int64_t* __restrict__ out_perm,
                                                              int32_t* __restrict__ out_counts) {
  __shared__ int lds[ODET_MAX_LEVELS * 17];
  int cnt = count_dev ? min(*count_dev, n) : n;
  d_assign_levels_block<AL_THREADS>(rois, cnt, min_level, max_level, out_rois, out_level, out_perm, out_counts, lds);
}

extern "C" int odet_assign_levels(const float* rois, int n, const int32_t* count_dev, int min_level, int max_level,
                                  float* out_rois, int32_t* out_level, int64_t* out_perm, int32_t* out_counts,
                                  odet_stream_t stream) {
  ODET_REQUIRE(n >= 0 && max_level >= min_level && max_level - min_level < ODET_MAX_LEVELS,
               "odet_assign_levels: bad sizes");
  ODET_REQUIRE(out_counts, "odet_assign_levels: null out_counts");
  if (n > ODET_ASSIGN_MAX_ROIS)
    return odet_set_error(ODET_E_LIMIT, "odet_assign_levels: n %d exceeds %d", n, ODET_ASSIGN_MAX_ROIS);
  if (n == 0) {
    ODET_HIP(hipMemsetAsync(out_counts, 0, sizeof(int32_t) * (max_level - min_level + 1), (hipStream_t)stream));
    return ODET_OK;
  }
  ODET_REQUIRE(rois && out_rois && out_level && out_perm, "odet_assign_levels: null pointer");
  hipLaunchKernelGGL(k_assign_levels, dim3(1), dim3(AL_THREADS), 0, (hipStream_t)stream, (const float4*)rois, n,
                     count_dev, min_level, max_level, (float4*)out_rois, out_level, out_perm, out_counts);
  ODET_LAUNCH_CHECK();
  return ODET_OK;
}

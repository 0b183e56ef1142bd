// Detection post-processing: model/prediction.py:103-163 post_ops_prediction.
//
// The reference runs a Python loop over the 20 (VOC) / 80 (COCO) foreground classes; each
// iteration dispatches ~12 TF ops, one single-threaded NMS and one device->host sync
// (prediction.py:147).  Here every class is one workgroup of ONE launch:
//
//   k_postops_class (grid = num_classes-1):
//     1. score filter (strict >, :136), decode (:138-140), clip + min-edge filter (:141-143):
//        one RoI per thread, box kept in LDS, 64-bit sort key (score desc, RoI index asc)
//     2. bitonic sort of the keys in LDS
//     3. wave 0: exact greedy NMS over the sorted candidates in 64-wide tiles -- each lane owns
//        a candidate; phase 1 tests the tile against the boxes already kept (LDS broadcast),
//        phase 2 resolves the tile with ctz over the alive ballot + v_readlane broadcast of the
//        winner's box; stops at max_per_class (:146).
//   k_postops_merge (1 workgroup): concatenation in class order (:156-158), top-k by
//     (score desc, position asc) (:160), gather (:162).
//
// Output order is the sorted order, one valid instance of tf.nn.top_k(sorted=False)'s
// unspecified order.
#include "odet_internal.h"

#define PO_THREADS 256

struct PostOpsParams {
  const float* scores;   // [R, Ccls]
  const float* deltas;   // [R, Ccls, 4]
  const float4* rois;    // [R]
  const int32_t* count_dev;
  int R, Ccls, P2, K;
  float means[4], stds[4];
  float wmax, hmax, min_edge, score_thr, nms_thr;
  // per-class results
  int32_t* cls_count;    // [ncls-1]
  float4* cls_boxes;     // [ncls-1, K]
  float* cls_scores;     // [ncls-1, K]
};

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

__global__ void __launch_bounds__(PO_THREADS) k_postops_class(PostOpsParams p) {
  extern __shared__ __align__(16) unsigned char smem[];
  unsigned long long* keys = reinterpret_cast<unsigned long long*>(smem);          // [P2]
  float4* lbox = reinterpret_cast<float4*>(smem + (size_t)p.P2 * 8);                // [R]
  float4* kbox = lbox + p.R;                                                        // [K]
  float* karea = reinterpret_cast<float*>(kbox + p.K);                              // [K]
  __shared__ int s_nvalid;

  const int c = blockIdx.x + 1;   // class id, prediction.py:135
  const int R = p.count_dev ? min(*p.count_dev, p.R) : p.R;
  if (threadIdx.x == 0) s_nvalid = 0;
  __syncthreads();

  // 1. filter + decode + clip
  for (int r = threadIdx.x; r < p.P2; r += PO_THREADS) {
    unsigned long long key = ~0ull;
    if (r < R) {
      float s = p.scores[(size_t)r * p.Ccls + c];
      if (s > p.score_thr) {                                                   // :136
        const float* t = p.deltas + ((size_t)r * p.Ccls + c) * 4;
        float d0 = t[0] * p.stds[0] + p.means[0];
        float d1 = t[1] * p.stds[1] + p.means[1];
        float d2 = t[2] * p.stds[2] + p.means[2];
        float d3 = t[3] * p.stds[3] + p.means[3];
        float4 b = d_decode_box(p.rois[r], d0, d1, d2, d3);                   // :138-140
        b = d_clip_box(b, 0.0f, p.wmax, p.hmax);                               // :141-143
        float e0 = b.z - b.x + 1.0f, e1 = b.w - b.y + 1.0f;                    // bbox_tf.py:81-83
        if (e1 >= p.min_edge && e0 >= p.min_edge) {
          lbox[r] = b;
          key = ((unsigned long long)(~d_float_asc_key(s)) << 32) | (unsigned)r;
          atomicAdd(&s_nvalid, 1);
        }
      }
    }
    keys[r] = key;
  }
  __syncthreads();

  // 2. sort: score desc, RoI index asc; rejected rows (key = ~0) go last
  bitonic_sort_u64(keys, p.P2, PO_THREADS);
  const int nvalid = s_nvalid;

  // 3. greedy NMS by wave 0
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    const int K = p.K;
    int nk = 0;
    for (int t0 = 0; t0 < nvalid && nk < K; t0 += 64) {
      const int ci = t0 + lane;
      const bool have = ci < nvalid;
      const int r = have ? (int)(keys[ci] & 0xFFFFFFFFull) : 0;
      const float4 ob = have ? lbox[r] : make_float4(0, 0, 0, 0);
      const float4 nb = d_norm_box(ob);
      const float area = d_box_area(nb);
      bool sup = false;
      for (int k = 0; k < nk; ++k) sup = sup || d_iou_gt(nb, area, kbox[k], karea[k], p.nms_thr);
      unsigned long long alive = __ballot(have && !sup);
      while (alive != 0 && nk < K) {
        const int i = __builtin_ctzll(alive);
        float4 wb;
        wb.x = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, nb.x), i));
        wb.y = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, nb.y), i));
        wb.z = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, nb.z), i));
        wb.w = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, nb.w), i));
        const float wa = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, area), i));
        const bool s2 = d_iou_gt(nb, area, wb, wa, p.nms_thr);
        alive &= ~__ballot(s2);
        alive &= ~(1ull << i);
        if (lane == i) {
          kbox[nk] = nb;
          karea[nk] = area;
          p.cls_boxes[(size_t)blockIdx.x * K + nk] = ob;
          p.cls_scores[(size_t)blockIdx.x * K + nk] = p.scores[(size_t)r * p.Ccls + c];
        }
        ++nk;
        // make the new kept box visible to the next tile's phase 1 (same wave, LDS in order)
      }
    }
    if (lane == 0) p.cls_count[blockIdx.x] = nk;
  }
}

struct MergeParams {
  const int32_t* cls_count;
  const float4* cls_boxes;
  const float* cls_scores;
  int ncls1, K, P2, max_per_image;
  float4* out_boxes;
  int32_t* out_labels;
  float* out_scores;
  int32_t* out_count;
};

__global__ void __launch_bounds__(1024) k_postops_merge(MergeParams p) {
  extern __shared__ __align__(16) unsigned char smem[];
  unsigned long long* keys = reinterpret_cast<unsigned long long*>(smem);   // [P2]
  uint32_t* src = reinterpret_cast<uint32_t*>(smem + (size_t)p.P2 * 8);      // [P2] slot of position p
  int* prefix = reinterpret_cast<int*>(src + p.P2);                          // [ncls1 + 1]
  if (threadIdx.x == 0) {
    int run = 0;
    for (int c = 0; c < p.ncls1; ++c) { prefix[c] = run; run += min(p.cls_count[c], p.K); }
    prefix[p.ncls1] = run;
  }
  for (int i = threadIdx.x; i < p.P2; i += 1024) keys[i] = ~0ull;
  __syncthreads();
  const int total = prefix[p.ncls1];
  // concatenation order: class ascending, NMS order inside a class (prediction.py:156-158)
  for (int slot = threadIdx.x; slot < p.ncls1 * p.K; slot += 1024) {
    int c = slot / p.K, k = slot - c * p.K;
    if (k < min(p.cls_count[c], p.K)) {
      int pos = prefix[c] + k;
      keys[pos] = ((unsigned long long)(~d_float_asc_key(p.cls_scores[slot])) << 32) | (unsigned)pos;
      src[pos] = (uint32_t)slot;
    }
  }
  __syncthreads();
  bitonic_sort_u64(keys, p.P2, 1024);
  const int M = min(total, p.max_per_image);                                 // prediction.py:160
  for (int i = threadIdx.x; i < M; i += 1024) {
    int pos = (int)(keys[i] & 0xFFFFFFFFull);
    uint32_t slot = src[pos];
    p.out_boxes[i] = p.cls_boxes[slot];
    p.out_scores[i] = p.cls_scores[slot];
    p.out_labels[i] = (int32_t)(slot / p.K) + 1;
  }
  if (threadIdx.x == 0) *p.out_count = M;
}

static int next_pow2(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}

extern "C" size_t odet_post_ops_workspace_bytes(int num_classes, int max_per_class) {
  size_t n1 = (size_t)(num_classes > 1 ? num_classes - 1 : 1);
  size_t k = (size_t)(max_per_class > 0 ? max_per_class : 1);
  return odet_align_up(n1 * 4, 256) + odet_align_up(n1 * k * 16, 256) + odet_align_up(n1 * k * 4, 256) + 1024;
}

extern "C" int odet_post_ops(const float* scores, const float* deltas, const float* rois, int R,
                             const int32_t* count_dev, int Ccls, int num_classes, int image_h, int image_w,
                             const float* means, const float* stds, int max_per_class, int max_per_image,
                             float nms_iou_threshold, float score_threshold, float min_edge, float* out_boxes,
                             int32_t* out_labels, float* out_scores, int32_t* out_count, void* workspace,
                             size_t workspace_bytes, odet_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  ODET_REQUIRE(out_count, "odet_post_ops: null out_count");
  ODET_REQUIRE(R >= 0 && Ccls > 0 && num_classes >= 1 && num_classes <= Ccls, "odet_post_ops: bad sizes");
  ODET_REQUIRE(max_per_class >= 0 && max_per_image >= 0, "odet_post_ops: negative cap");
  if (R == 0 || num_classes == 1 || max_per_class == 0 || max_per_image == 0) {
    ODET_HIP(hipMemsetAsync(out_count, 0, sizeof(int32_t), st));
    return ODET_OK;
  }
  ODET_REQUIRE(scores && deltas && rois && means && stds && out_boxes && out_labels && out_scores,
               "odet_post_ops: null pointer");
  if (R > ODET_POSTOPS_MAX_ROIS)
    return odet_set_error(ODET_E_LIMIT, "odet_post_ops: R %d exceeds %d", R, ODET_POSTOPS_MAX_ROIS);
  const int ncls1 = num_classes - 1;
  if ((int64_t)ncls1 * max_per_class > ODET_POSTOPS_MAX_CANDIDATES)
    return odet_set_error(ODET_E_LIMIT, "odet_post_ops: (num_classes-1)*max_per_class %lld exceeds %d",
                          (long long)ncls1 * max_per_class, ODET_POSTOPS_MAX_CANDIDATES);
  size_t need = odet_post_ops_workspace_bytes(num_classes, max_per_class);
  if (!workspace || workspace_bytes < need)
    return odet_set_error(ODET_E_WORKSPACE, "odet_post_ops: workspace too small (%zu < %zu)", workspace_bytes, need);
  OdetArena ar{(char*)workspace, workspace_bytes, 0};
  PostOpsParams p;
  p.scores = scores; p.deltas = deltas; p.rois = (const float4*)rois; p.count_dev = count_dev;
  p.R = R; p.Ccls = Ccls; p.P2 = next_pow2(R < 2 ? 2 : R); p.K = max_per_class;
  for (int k = 0; k < 4; ++k) { p.means[k] = means[k]; p.stds[k] = stds[k]; }
  p.wmax = (float)(image_w - 1); p.hmax = (float)(image_h - 1);
  p.min_edge = min_edge; p.score_thr = score_threshold; p.nms_thr = nms_iou_threshold;
  p.cls_count = ar.take<int32_t>(ncls1);
  p.cls_boxes = ar.take<float4>((size_t)ncls1 * max_per_class);
  p.cls_scores = ar.take<float>((size_t)ncls1 * max_per_class);
  size_t lds1 = (size_t)p.P2 * 8 + (size_t)R * 16 + (size_t)max_per_class * 20;
  if (lds1 > 150 * 1024)
    return odet_set_error(ODET_E_LIMIT, "odet_post_ops: R/max_per_class need %zu B of LDS (> 150 KiB)", lds1);
  static bool attr_set = false;
  if (!attr_set) {
    ODET_HIP(hipFuncSetAttribute((const void*)k_postops_class, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    ODET_HIP(hipFuncSetAttribute((const void*)k_postops_merge, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL(k_postops_class, dim3(ncls1), dim3(PO_THREADS), lds1, st, p);
  ODET_LAUNCH_CHECK();
  MergeParams m;
  m.cls_count = p.cls_count; m.cls_boxes = p.cls_boxes; m.cls_scores = p.cls_scores;
  m.ncls1 = ncls1; m.K = max_per_class; m.P2 = next_pow2(ncls1 * max_per_class < 2 ? 2 : ncls1 * max_per_class);
  m.max_per_image = max_per_image;
  m.out_boxes = (float4*)out_boxes; m.out_labels = out_labels; m.out_scores = out_scores; m.out_count = out_count;
  size_t lds2 = (size_t)m.P2 * 12 + (size_t)(ncls1 + 1) * 4;
  hipLaunchKernelGGL(k_postops_merge, dim3(1), dim3(1024), lds2, st, m);
  ODET_LAUNCH_CHECK();
  return ODET_OK;
}

// ---------------------------------------------------------------------- detection records --
// Fixed-size record of one image for the image-parallel all-gather: float32 [max_det, 6] rows
// (x1, y1, x2, y2, score, label), padded rows are zero with score = -1, then one float holding
// the count.  One launch, no host sync.
__global__ void __launch_bounds__(256) k_pack_detections(const float4* __restrict__ boxes,
                                                         const int32_t* __restrict__ labels,
                                                         const float* __restrict__ scores,
                                                         const int32_t* __restrict__ count, int cap, int max_det,
                                                         float* __restrict__ rec) {
  const int m = min(min(*count, cap), max_det);
  for (int i = threadIdx.x; i < max_det; i += 256) {
    float* r = rec + (size_t)i * 6;
    if (i < m) {
      float4 b = boxes[i];
      r[0] = b.x; r[1] = b.y; r[2] = b.z; r[3] = b.w; r[4] = scores[i]; r[5] = (float)labels[i];
    } else {
      r[0] = 0.0f; r[1] = 0.0f; r[2] = 0.0f; r[3] = 0.0f; r[4] = -1.0f; r[5] = 0.0f;
    }
  }
  if (threadIdx.x == 0) rec[(size_t)max_det * 6] = (float)m;
}

extern "C" int odet_pack_detections(const float* boxes, const int32_t* labels, const float* scores,
                                    const int32_t* count_dev, int capacity, int max_det, float* out_record,
                                    odet_stream_t stream) {
  ODET_REQUIRE(boxes && labels && scores && count_dev && out_record, "odet_pack_detections: null pointer");
  ODET_REQUIRE(capacity >= 0 && max_det > 0, "odet_pack_detections: bad sizes");
  hipLaunchKernelGGL(k_pack_detections, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float4*)boxes, labels,
                     scores, count_dev, capacity, max_det, out_record);
  ODET_LAUNCH_CHECK();
  return ODET_OK;
}

/* CPU ORACLE (plain C restatement) -- TEST INFRASTRUCTURE ONLY, NOT PRODUCT CODE.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load or
 * call this file (through oracle/c_oracle.py), and only as the checker / the reported CPU
 * baseline.  The product library (tf_eager_object_detection_amd/csrc -> libodet_hip.so)
 * shares no code with it.
 *
 * It restates, in the reference's operation order and in float32, the Faster-R-CNN / FPN
 * inference hot path of irvingzhang0512/tf_eager_object_detection and the TensorFlow 1.x
 * CPU kernels it delegates to.  Each function cites the reference file:line it follows
 * (paths relative to the reference checkout; "TF r1.13" = the third-party kernel, restated
 * from its published source because TensorFlow is neither vendored by the reference nor
 * installable here -> PARITY UNPINNED for those parts, see oracle/oracle_np.py header).
 *
 * It is written "as the reference runs it": priority-queue greedy NMS over ALL anchors,
 * un-fused crop_and_resize 14x14 -> materialised intermediate -> separate 2x2 max-pool,
 * sequential per-class loop.  That is what bench.py times as the CPU baseline
 * (kind = "port").
 *
 * Build: gcc -O2 -fPIC -shared -ffp-contract=off -fno-fast-math -fopenmp oracle.c -lm
 * exp/log are the correctly rounded float32 functions ((float)exp((double)x)).
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static inline float exp32(float x) { return (float)exp((double)x); }
static inline float log32(float x) { return (float)log((double)x); }
static inline float fmin32(float a, float b) { return a < b ? a : b; }
static inline float fmax32(float a, float b) { return a > b ? a : b; }

/* ---------------------------------------------------------------- anchors ------------ */

/* utils/anchor_generator.py:46-60 generate_by_anchor_base_tf: int32 shifts, row-major
 * cells (y outer, x inner), anchor-minor, float32 add. */
void orc_anchors_shift(const float* base, int A, int stride, int fh, int fw, float* out) {
  for (int y = 0; y < fh; ++y)
    for (int x = 0; x < fw; ++x) {
      float sx = (float)(x * stride), sy = (float)(y * stride);
      float* o = out + ((size_t)(y * fw + x) * A) * 4;
      for (int a = 0; a < A; ++a) {
        o[a * 4 + 0] = base[a * 4 + 0] + sx;
        o[a * 4 + 1] = base[a * 4 + 1] + sy;
        o[a * 4 + 2] = base[a * 4 + 2] + sx;
        o[a * 4 + 3] = base[a * 4 + 3] + sy;
      }
    }
}

/* utils/anchor_generator.py:137-178 make_anchors (+enum_scales :165-167, enum_ratios
 * :170-178 incl. the (hs, ws) swap at :178/:143): w = S*s*sqrt(r), h = (S*s)/sqrt(r),
 * ratio-major / scale-minor, location-major / anchor-minor. */
void orc_make_anchors(float base_size, const float* scales, int ns, const float* ratios, int nr,
                      int fh, int fw, int stride, float* out) {
  int A = ns * nr;
  float* w = (float*)malloc(sizeof(float) * A);
  float* h = (float*)malloc(sizeof(float) * A);
  for (int r = 0; r < nr; ++r) {
    float sr = sqrtf(ratios[r]);
    for (int s = 0; s < ns; ++s) {
      float side = base_size * scales[s];
      w[r * ns + s] = side * sr;   /* "hs" inside enum_ratios, received as ws */
      h[r * ns + s] = side / sr;   /* "ws" inside enum_ratios, received as hs */
    }
  }
  float fs = (float)stride;
  for (int y = 0; y < fh; ++y)
    for (int x = 0; x < fw; ++x) {
      float cx = (float)x * fs, cy = (float)y * fs;
      float* o = out + ((size_t)(y * fw + x) * A) * 4;
      for (int a = 0; a < A; ++a) {
        float hw = 0.5f * w[a], hh = 0.5f * h[a];
        o[a * 4 + 0] = cx - hw;
        o[a * 4 + 1] = cy - hh;
        o[a * 4 + 2] = cx + hw;
        o[a * 4 + 3] = cy + hh;
      }
    }
  free(w);
  free(h);
}

/* ---------------------------------------------------------------- box transforms ----- */

/* utils/bbox_transform.py:32-55 */
void orc_decode(const float* anchors, const float* deltas, int n, const float* means,
                const float* stds, float* out) {
  for (int i = 0; i < n; ++i) {
    const float* a = anchors + (size_t)i * 4;
    const float* t = deltas + (size_t)i * 4;
    float d0 = t[0] * stds[0] + means[0];
    float d1 = t[1] * stds[1] + means[1];
    float d2 = t[2] * stds[2] + means[2];
    float d3 = t[3] * stds[3] + means[3];
    float width = a[2] - a[0] + 1.0f;
    float height = a[3] - a[1] + 1.0f;
    float cx = a[0] + 0.5f * width;
    float cy = a[1] + 0.5f * height;
    cx = cx + d0 * width;
    cy = cy + d1 * height;
    width = width * exp32(d2);
    height = height * exp32(d3);
    float x1 = cx - 0.5f * width;
    float y1 = cy - 0.5f * height;
    out[(size_t)i * 4 + 0] = x1;
    out[(size_t)i * 4 + 1] = y1;
    out[(size_t)i * 4 + 2] = x1 + width;
    out[(size_t)i * 4 + 3] = y1 + height;
  }
}

/* utils/bbox_transform.py:4-29 */
void orc_encode(const float* src, const float* dst, int n, const float* means, const float* stds,
                float* out) {
  for (int i = 0; i < n; ++i) {
    const float* b = src + (size_t)i * 4;
    const float* g = dst + (size_t)i * 4;
    float width = b[2] - b[0] + 1.0f, height = b[3] - b[1] + 1.0f;
    float cx = b[0] + 0.5f * width, cy = b[1] + 0.5f * height;
    float gw = g[2] - g[0] + 1.0f, gh = g[3] - g[1] + 1.0f;
    float gcx = g[0] + 0.5f * gw, gcy = g[1] + 0.5f * gh;
    float d[4];
    d[0] = (gcx - cx) / width;
    d[1] = (gcy - cy) / height;
    d[2] = log32(gw / width);
    d[3] = log32(gh / height);
    for (int k = 0; k < 4; ++k) out[(size_t)i * 4 + k] = (d[k] - means[k]) / stds[k];
  }
}

/* utils/bbox_tf.py:59-84.  min_edge < 0 means "None": clip only, idx = 0..n-1.
 * Returns the number of rows kept. */
int orc_clip_filter(const float* in, int n, float min_value, int max_h, int max_w, float min_edge,
                    float* out, int64_t* idx) {
  float wm = (float)(max_w - 1), hm = (float)(max_h - 1);
  int m = 0;
  for (int i = 0; i < n; ++i) {
    const float* b = in + (size_t)i * 4;
    float c0 = fmax32(fmin32(b[0], wm), min_value);
    float c1 = fmax32(fmin32(b[1], hm), min_value);
    float c2 = fmax32(fmin32(b[2], wm), min_value);
    float c3 = fmax32(fmin32(b[3], hm), min_value);
    if (min_edge >= 0.0f) {
      float e0 = c2 - c0 + 1.0f, e1 = c3 - c1 + 1.0f;
      if (!(e1 >= min_edge && e0 >= min_edge)) continue;
    }
    out[(size_t)m * 4 + 0] = c0;
    out[(size_t)m * 4 + 1] = c1;
    out[(size_t)m * 4 + 2] = c2;
    out[(size_t)m * 4 + 3] = c3;
    if (idx) idx[m] = i;
    ++m;
  }
  return m;
}

/* utils/bbox_tf.py:87-101 */
int orc_range_filter(const float* a, int n, int max_h, int max_w, int64_t* idx) {
  float wm = (float)(max_w - 1), hm = (float)(max_h - 1);
  int m = 0;
  for (int i = 0; i < n; ++i) {
    const float* b = a + (size_t)i * 4;
    if (b[0] >= 0 && b[1] >= 0 && b[2] <= wm && b[3] <= hm) idx[m++] = i;
  }
  return m;
}

/* utils/bbox_tf.py:7-56 (area / pairwise_intersection / pairwise_iou, +1 convention) */
void orc_pairwise_iou(const float* b1, int n, const float* b2, int m, float* out) {
  for (int i = 0; i < n; ++i) {
    const float* p = b1 + (size_t)i * 4;
    float a1 = (p[3] - p[1] + 1.0f) * (p[2] - p[0] + 1.0f);
    for (int j = 0; j < m; ++j) {
      const float* q = b2 + (size_t)j * 4;
      float a2 = (q[3] - q[1] + 1.0f) * (q[2] - q[0] + 1.0f);
      float ih = fmax32(0.0f, fmin32(p[3], q[3]) - fmax32(p[1], q[1]) + 1.0f);
      float iw = fmax32(0.0f, fmin32(p[2], q[2]) - fmax32(p[0], q[0]) + 1.0f);
      float inter = ih * iw;
      float uni = a1 + a2 - inter;
      out[(size_t)i * m + j] = (inter == 0.0f) ? 0.0f : inter / uni;
    }
  }
}

/* ---------------------------------------------------------------- TF NMS ------------- */

/* TF r1.13 non_max_suppression_op.cc: IOUGreaterThanThreshold */
static inline int tf_iou_gt(const float* bi, const float* bj, float thr) {
  float ymin_i = fmin32(bi[0], bi[2]), xmin_i = fmin32(bi[1], bi[3]);
  float ymax_i = fmax32(bi[0], bi[2]), xmax_i = fmax32(bi[1], bi[3]);
  float ymin_j = fmin32(bj[0], bj[2]), xmin_j = fmin32(bj[1], bj[3]);
  float ymax_j = fmax32(bj[0], bj[2]), xmax_j = fmax32(bj[1], bj[3]);
  float area_i = (ymax_i - ymin_i) * (xmax_i - xmin_i);
  float area_j = (ymax_j - ymin_j) * (xmax_j - xmin_j);
  if (area_i <= 0 || area_j <= 0) return 0;
  float iymin = fmax32(ymin_i, ymin_j), ixmin = fmax32(xmin_i, xmin_j);
  float iymax = fmin32(ymax_i, ymax_j), ixmax = fmin32(xmax_i, xmax_j);
  float inter = fmax32(iymax - iymin, 0.0f) * fmax32(ixmax - ixmin, 0.0f);
  float iou = inter / (area_i + area_j - inter);
  return iou > thr;
}

typedef struct { float score; int32_t idx; } cand_t;
/* heap order: a "less" than b  <=>  lower score, or equal score and larger index */
static inline int cand_less(cand_t a, cand_t b) {
  return a.score < b.score || (a.score == b.score && a.idx > b.idx);
}
static void heap_push(cand_t* h, int* n, cand_t c) {
  int i = (*n)++;
  h[i] = c;
  while (i > 0) {
    int p = (i - 1) / 2;
    if (!cand_less(h[p], h[i])) break;
    cand_t t = h[p]; h[p] = h[i]; h[i] = t;
    i = p;
  }
}
static cand_t heap_pop(cand_t* h, int* n) {
  cand_t top = h[0];
  int m = --(*n);
  h[0] = h[m];
  int i = 0;
  for (;;) {
    int l = 2 * i + 1, r = l + 1, b = i;
    if (l < m && cand_less(h[b], h[l])) b = l;
    if (r < m && cand_less(h[b], h[r])) b = r;
    if (b == i) break;
    cand_t t = h[b]; h[b] = h[i]; h[i] = t;
    i = b;
  }
  return top;
}

/* tf.image.non_max_suppression = NonMaxSuppressionV3(score_threshold = lowest float)
 * (call sites model/region_proposal.py:74-76, model/prediction.py:146).
 * stats (optional, int64[2]): [0] = candidates popped, [1] = IoU evaluations. */
int orc_nms(const float* boxes, const float* scores, int n, int max_out, float thr,
            int32_t* out_idx, int64_t* stats) {
  int out_size = max_out < n ? max_out : n;
  cand_t* heap = (cand_t*)malloc(sizeof(cand_t) * (size_t)(n > 0 ? n : 1));
  int hn = 0;
  const float lowest = -3.402823466e+38f;
  for (int i = 0; i < n; ++i)
    if (scores[i] > lowest) { cand_t c = {scores[i], i}; heap_push(heap, &hn, c); }
  int kept = 0;
  int64_t pops = 0, ious = 0;
  while (kept < out_size && hn > 0) {
    cand_t c = heap_pop(heap, &hn);
    ++pops;
    int keep = 1;
    const float* bi = boxes + (size_t)c.idx * 4;
    for (int j = kept - 1; j >= 0; --j) {
      ++ious;
      if (tf_iou_gt(bi, boxes + (size_t)out_idx[j] * 4, thr)) { keep = 0; break; }
    }
    if (keep) out_idx[kept++] = c.idx;
  }
  free(heap);
  if (stats) { stats[0] = pops; stats[1] = ious; }
  return kept;
}

/* ---------------------------------------------------------------- TF crop_and_resize - */

/* TF r1.13 crop_and_resize_op.cc CPU functor, bilinear, extrapolation_value = 0
 * (call sites model/roi_pooling.py:37,79,86,134).  boxes normalised (y1,x1,y2,x2);
 * batch index is always 0 in the reference (roi_pooling.py:28,66,152).  TF shards the
 * boxes over its intra-op pool -> `threads` (OpenMP). */
void orc_crop_and_resize(const float* img, int H, int W, int C, const float* boxes, int R,
                         int ch, int cw, float* out, int threads) {
  const float Hm1 = (float)(H - 1), Wm1 = (float)(W - 1);
#ifdef _OPENMP
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(dynamic, 4)
#endif
  for (int b = 0; b < R; ++b) {
    const float y1 = boxes[b * 4 + 0], x1 = boxes[b * 4 + 1];
    const float y2 = boxes[b * 4 + 2], x2 = boxes[b * 4 + 3];
    const float hs = (ch > 1) ? (y2 - y1) * Hm1 / (float)(ch - 1) : 0.0f;
    const float ws = (cw > 1) ? (x2 - x1) * Wm1 / (float)(cw - 1) : 0.0f;
    for (int y = 0; y < ch; ++y) {
      float* orow = out + (((size_t)b * ch + y) * cw) * C;
      const float in_y = (ch > 1) ? y1 * Hm1 + (float)y * hs : 0.5f * (y1 + y2) * Hm1;
      if (in_y < 0 || in_y > Hm1) { memset(orow, 0, sizeof(float) * (size_t)cw * C); continue; }
      const int top = (int)floorf(in_y), bot = (int)ceilf(in_y);
      const float y_lerp = in_y - (float)top;
      for (int x = 0; x < cw; ++x) {
        float* o = orow + (size_t)x * C;
        const float in_x = (cw > 1) ? x1 * Wm1 + (float)x * ws : 0.5f * (x1 + x2) * Wm1;
        if (in_x < 0 || in_x > Wm1) { memset(o, 0, sizeof(float) * (size_t)C); continue; }
        const int left = (int)floorf(in_x), right = (int)ceilf(in_x);
        const float x_lerp = in_x - (float)left;
        const float* tl = img + ((size_t)top * W + left) * C;
        const float* tr = img + ((size_t)top * W + right) * C;
        const float* bl = img + ((size_t)bot * W + left) * C;
        const float* br = img + ((size_t)bot * W + right) * C;
        for (int d = 0; d < C; ++d) {
          const float t = tl[d] + (tr[d] - tl[d]) * x_lerp;
          const float bt = bl[d] + (br[d] - bl[d]) * x_lerp;
          o[d] = t + (bt - t) * y_lerp;
        }
      }
    }
  }
}

/* Keras MaxPooling2D(2x2/2,'same') on an even map (model/roi_pooling.py:13,42,51,84) */
void orc_max_pool2(const float* in, int R, int H, int W, int C, float* out, int threads) {
  int Ho = H / 2, Wo = W / 2;
#ifdef _OPENMP
#pragma omp parallel for num_threads(threads > 0 ? threads : 1)
#endif
  for (int b = 0; b < R; ++b)
    for (int y = 0; y < Ho; ++y)
      for (int x = 0; x < Wo; ++x) {
        const float* p00 = in + ((((size_t)b * H + 2 * y) * W) + 2 * x) * C;
        const float* p01 = p00 + C;
        const float* p10 = p00 + (size_t)W * C;
        const float* p11 = p10 + C;
        float* o = out + ((((size_t)b * Ho + y) * Wo) + x) * C;
        for (int d = 0; d < C; ++d) o[d] = fmax32(fmax32(p00[d], p01[d]), fmax32(p10[d], p11[d]));
      }
}

/* tf.nn.avg_pool 2x2/2 'SAME' on an even map (model/roi_pooling.py:154): row-major window
 * sum, then / 4 */
void orc_avg_pool2(const float* in, int R, int H, int W, int C, float* out, int threads) {
  int Ho = H / 2, Wo = W / 2;
#ifdef _OPENMP
#pragma omp parallel for num_threads(threads > 0 ? threads : 1)
#endif
  for (int b = 0; b < R; ++b)
    for (int y = 0; y < Ho; ++y)
      for (int x = 0; x < Wo; ++x) {
        const float* p00 = in + ((((size_t)b * H + 2 * y) * W) + 2 * x) * C;
        const float* p01 = p00 + C;
        const float* p10 = p00 + (size_t)W * C;
        const float* p11 = p10 + C;
        float* o = out + ((((size_t)b * Ho + y) * Wo) + x) * C;
        for (int d = 0; d < C; ++d) o[d] = (((p00[d] + p01[d]) + p10[d]) + p11[d]) / 4.0f;
      }
}

/* model/roi_pooling.py:53-90 RoiPoolingCropAndResize.call (stride variant) when
 * image_h <= 0, model/roi_pooling.py:15-42 RoiPoolingCropAndResize2.call (FPN variant,
 * normalise by image size) when image_h > 0.  Un-fused: materialises the 2P x 2P crops
 * in `scratch` (R*2P*2P*C floats) and then pools, as the reference does. */
void orc_roi_pool(const float* feat, int H, int W, int C, const float* rois, int R, float stride,
                  int image_h, int image_w, int pool, int max_pool_flag, float* out,
                  float* scratch, int threads) {
  float* nb = (float*)malloc(sizeof(float) * 4 * (size_t)(R > 0 ? R : 1));
  for (int r = 0; r < R; ++r) {
    const float* b = rois + (size_t)r * 4;
    if (image_h > 0) {
      float h = (float)image_h, w = (float)image_w;
      nb[r * 4 + 0] = b[1] / h; nb[r * 4 + 1] = b[0] / w;
      nb[r * 4 + 2] = b[3] / h; nb[r * 4 + 3] = b[2] / w;
    } else {
      float x1 = b[0] / stride, y1 = b[1] / stride, x2 = b[2] / stride, y2 = b[3] / stride;
      float hm = (float)(H - 1), wm = (float)(W - 1);
      nb[r * 4 + 0] = y1 / hm; nb[r * 4 + 1] = x1 / wm;
      nb[r * 4 + 2] = y2 / hm; nb[r * 4 + 3] = x2 / wm;
    }
  }
  if (max_pool_flag) {
    orc_crop_and_resize(feat, H, W, C, nb, R, 2 * pool, 2 * pool, scratch, threads);
    orc_max_pool2(scratch, R, 2 * pool, 2 * pool, C, out, threads);
  } else {
    orc_crop_and_resize(feat, H, W, C, nb, R, pool, pool, out, threads);
  }
  free(nb);
}

/* model/roi_pooling.py:93-177: crop_and_resize(pad_border=True) + roi_align +
 * RoiPoolingRoiAlign.call.  scratch: R*2P*2P*C floats; padded: (H+2)*(W+2)*C floats. */
void orc_roi_align(const float* feat, int H, int W, int C, const float* rois, int R, float stride,
                   int pool, float* out, float* scratch, float* padded, int threads) {
  int Hp = H + 2, Wp = W + 2, crop = 2 * pool;
  for (int y = 0; y < Hp; ++y) {          /* tf.pad SYMMETRIC 1px == edge replicate (:100) */
    int sy = y == 0 ? 0 : (y == Hp - 1 ? H - 1 : y - 1);
    for (int x = 0; x < Wp; ++x) {
      int sx = x == 0 ? 0 : (x == Wp - 1 ? W - 1 : x - 1);
      memcpy(padded + ((size_t)y * Wp + x) * C, feat + ((size_t)sy * W + sx) * C, sizeof(float) * C);
    }
  }
  float* nb = (float*)malloc(sizeof(float) * 4 * (size_t)(R > 0 ? R : 1));
  float cs = (float)crop, imh = (float)(Hp - 1), imw = (float)(Wp - 1);
  for (int r = 0; r < R; ++r) {
    const float* b = rois + (size_t)r * 4;
    float x0 = b[0] / stride + 1.0f, y0 = b[1] / stride + 1.0f;      /* :175, :101 */
    float x1 = b[2] / stride + 1.0f, y1 = b[3] / stride + 1.0f;
    float sw = (x1 - x0) / cs, sh = (y1 - y0) / cs;                  /* :120-121 */
    float nx0 = (x0 + sw / 2.0f - 0.5f) / imw;                       /* :124 */
    float ny0 = (y0 + sh / 2.0f - 0.5f) / imh;
    float nw = sw * (float)(crop - 1) / imw;                         /* :127 */
    float nh = sh * (float)(crop - 1) / imh;
    nb[r * 4 + 0] = ny0; nb[r * 4 + 1] = nx0; nb[r * 4 + 2] = ny0 + nh; nb[r * 4 + 3] = nx0 + nw;
  }
  orc_crop_and_resize(padded, Hp, Wp, C, nb, R, crop, crop, scratch, threads);
  orc_avg_pool2(scratch, R, crop, crop, C, out, threads);
  free(nb);
}

/* ---------------------------------------------------------------- softmax / glue ----- */

/* tf.nn.softmax CPU (softmax_op_functor.h): exp(x - max) * (1 / sum) */
void orc_softmax(const float* logits, int rows, int cols, float* out) {
  for (int r = 0; r < rows; ++r) {
    const float* x = logits + (size_t)r * cols;
    float* o = out + (size_t)r * cols;
    float m = x[0];
    for (int c = 1; c < cols; ++c) m = fmax32(m, x[c]);
    float s = 0.0f;
    for (int c = 0; c < cols; ++c) { o[c] = exp32(x[c] - m); s = (c == 0) ? o[c] : s + o[c]; }
    float inv = 1.0f / s;
    for (int c = 0; c < cols; ++c) o[c] = o[c] * inv;
  }
}

/* model/fpn/base_fpn_model.py:223 (+:429): logits [n,2] interleaved (bg,fg) -> fg prob */
void orc_rpn_fg_fpn(const float* logits, int n, float* out) {
  for (int i = 0; i < n; ++i) {
    float a = logits[2 * i], b = logits[2 * i + 1];
    float m = fmax32(a, b);
    float e0 = exp32(a - m), e1 = exp32(b - m);
    float inv = 1.0f / (e0 + e1);
    out[i] = e1 * inv;
  }
}

/* model/faster_rcnn/base_faster_rcnn_model.py:149-152: per location channels [A bg | A fg] */
void orc_rpn_fg_frcnn(const float* logits, int nloc, int A, float* out) {
  for (int l = 0; l < nloc; ++l)
    for (int a = 0; a < A; ++a) {
      float x0 = logits[(size_t)l * 2 * A + a], x1 = logits[(size_t)l * 2 * A + A + a];
      float m = fmax32(x0, x1);
      float e0 = exp32(x0 - m), e1 = exp32(x1 - m);
      float inv = 1.0f / (e0 + e1);
      out[(size_t)l * A + a] = e1 * inv;
    }
}

/* model/fpn/base_fpn_model.py:303-324 _assign_levels.  levels[i] in [minl,maxl];
 * perm = concat over levels of ascending indices; counts[maxl-minl+1]. */
void orc_assign_levels(const float* rois, int R, int minl, int maxl, int32_t* levels,
                       int64_t* perm, int32_t* counts) {
  const float log2f32 = log32(2.0f);
  for (int i = 0; i < R; ++i) {
    const float* b = rois + (size_t)i * 4;
    float h = fmax32(0.0f, b[3] - b[1]);
    float w = fmax32(0.0f, b[2] - b[0]);
    float lv = floorf(4.0f + log32(sqrtf(w * h + 1e-8f) / 224.0f) / log2f32);
    lv = fmax32(lv, (float)minl);
    lv = fmin32(lv, (float)maxl);
    levels[i] = (int32_t)lv;
  }
  int m = 0;
  for (int l = minl; l <= maxl; ++l) {
    int c = 0;
    for (int i = 0; i < R; ++i)
      if (levels[i] == l) { perm[m++] = i; ++c; }
    counts[l - minl] = c;
  }
}

/* ---------------------------------------------------------------- region proposal ---- */

/* model/region_proposal.py:55-81: decode -> clip (no filter) -> NMS over ALL n -> gather.
 * `boxes_scratch`: n*4 floats.  Returns K' (<= K). */
int orc_region_proposal(const float* deltas, const float* anchors, const float* scores, int n,
                        int H, int W, const float* means, const float* stds, int K, float thr,
                        float* boxes_scratch, float* out_rois, int32_t* out_idx, int64_t* stats) {
  orc_decode(anchors, deltas, n, means, stds, boxes_scratch);
  orc_clip_filter(boxes_scratch, n, 0.0f, H, W, -1.0f, boxes_scratch, NULL);
  int k = orc_nms(boxes_scratch, scores, n, K, thr, out_idx, stats);
  for (int i = 0; i < k; ++i) memcpy(out_rois + (size_t)i * 4, boxes_scratch + (size_t)out_idx[i] * 4, 16);
  return k;
}

/* ---------------------------------------------------------------- post-processing ---- */

static int cmp_topk(const void* pa, const void* pb, void* ctx) {
  const float* v = (const float*)ctx;
  int a = *(const int*)pa, b = *(const int*)pb;
  if (v[a] > v[b]) return -1;
  if (v[a] < v[b]) return 1;
  return a < b ? -1 : (a > b ? 1 : 0);
}

/* model/prediction.py:103-163 post_ops_prediction.  S [R,Ccls], D [R,Ccls,4], rois [R,4].
 * Loops classes 1..num_classes-1 sequentially (:135), per class: score filter (strict >),
 * decode, clip + min_edge filter, NMS; concatenation class-ascending; final
 * top_k(min(max_per_image, n)) in (score desc, position asc) order.
 * Outputs sized max_per_image.  Returns M (0 => the reference returns (None,None,None)). */
int orc_post_ops(const float* S, const float* D, const float* rois, int R, int Ccls, int H, int W,
                 const float* means, const float* stds, int max_per_class, int max_per_image,
                 float nms_thr, float score_thr, float min_edge, int num_classes,
                 float* out_boxes, int32_t* out_cls, float* out_scores) {
  size_t cap = (size_t)(num_classes > 1 ? num_classes - 1 : 1) * (size_t)(max_per_class > 0 ? max_per_class : 1);
  float* all_b = (float*)malloc(sizeof(float) * 4 * cap);
  float* all_s = (float*)malloc(sizeof(float) * cap);
  int32_t* all_c = (int32_t*)malloc(sizeof(int32_t) * cap);
  size_t Rn = (size_t)(R > 0 ? R : 1);
  float* g_rois = (float*)malloc(sizeof(float) * 4 * Rn);
  float* g_del = (float*)malloc(sizeof(float) * 4 * Rn);
  float* g_sc = (float*)malloc(sizeof(float) * Rn);
  float* dec = (float*)malloc(sizeof(float) * 4 * Rn);
  float* clipped = (float*)malloc(sizeof(float) * 4 * Rn);
  float* sc2 = (float*)malloc(sizeof(float) * Rn);
  int64_t* sel = (int64_t*)malloc(sizeof(int64_t) * Rn);
  int32_t* keep = (int32_t*)malloc(sizeof(int32_t) * (size_t)(max_per_class > 0 ? max_per_class : 1));
  int total = 0;
  for (int c = 1; c < num_classes; ++c) {
    int m = 0;
    for (int r = 0; r < R; ++r) {
      float s = S[(size_t)r * Ccls + c];
      if (s > score_thr) {
        memcpy(g_rois + (size_t)m * 4, rois + (size_t)r * 4, 16);
        memcpy(g_del + (size_t)m * 4, D + ((size_t)r * Ccls + c) * 4, 16);
        g_sc[m++] = s;
      }
    }
    orc_decode(g_rois, g_del, m, means, stds, dec);
    int m2 = orc_clip_filter(dec, m, 0.0f, H, W, min_edge, clipped, sel);
    for (int i = 0; i < m2; ++i) sc2[i] = g_sc[sel[i]];
    int k = orc_nms(clipped, sc2, m2, max_per_class, nms_thr, keep, NULL);
    for (int i = 0; i < k; ++i) {
      memcpy(all_b + (size_t)total * 4, clipped + (size_t)keep[i] * 4, 16);
      all_s[total] = sc2[keep[i]];
      all_c[total] = c;
      ++total;
    }
  }
  int M = total < max_per_image ? total : max_per_image;
  if (total > 0) {
    int* order = (int*)malloc(sizeof(int) * (size_t)total);
    for (int i = 0; i < total; ++i) order[i] = i;
    qsort_r(order, (size_t)total, sizeof(int), cmp_topk, all_s);
    for (int i = 0; i < M; ++i) {
      memcpy(out_boxes + (size_t)i * 4, all_b + (size_t)order[i] * 4, 16);
      out_cls[i] = all_c[order[i]];
      out_scores[i] = all_s[order[i]];
    }
    free(order);
  }
  free(all_b); free(all_s); free(all_c); free(g_rois); free(g_del); free(g_sc);
  free(dec); free(clipped); free(sc2); free(sel); free(keep);
  return M;
}

int orc_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

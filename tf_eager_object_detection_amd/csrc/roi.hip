// RoI feature extraction: tf.image.crop_and_resize (bilinear, extrapolation 0) fused with the
// reference's 2x2 max / avg pooling, over all pyramid levels in one launch.
//
//   model/roi_pooling.py:45-90   RoiPoolingCropAndResize   (NORM_STRIDE, POOL_MAX2 | POOL_NONE)
//   model/roi_pooling.py:8-42    RoiPoolingCropAndResize2  (NORM_IMAGE,  POOL_MAX2)   <- FPN
//   model/roi_pooling.py:93-177  crop_and_resize/roi_align/RoiPoolingRoiAlign
//                                                          (NORM_TP_ALIGN, POOL_AVG2)
//
// The reference materialises the [R,14,14,C] crops (200 MB at R=1000, C=256) and pools them in
// a second op.  Here one wave produces one output bin: it loads the <=16 feature cells its 2x2
// samples tap (NHWC: a cell's C channels are contiguous, 64 lanes x float4 = 1 KiB coalesced per
// cell), lerps in the exact TF operation order (no FMA) and reduces in registers -- max is exact
// and the avg uses the same row-major sum, so results are bit-identical to the un-fused form.
//
// Work decomposition: task = (roi, py, px) -> one wave; 4 waves per workgroup; consecutive
// tasks of consecutive (level-sorted) RoIs are mapped to the same XCD so that each XCD's L2
// mostly holds one pyramid level.
#include "odet_internal.h"

struct RoiParams {
  const float* data[ODET_MAX_LEVELS];
  int H[ODET_MAX_LEVELS];
  int W[ODET_MAX_LEVELS];
  float stride[ODET_MAX_LEVELS];
  int C, n, norm_mode, P, pool_mode, num_levels;
  float image_h, image_w;
  int nblocks;        // logical workgroups (before padding the grid to a multiple of 8)
  int blocks_per_xcd;
};

struct Axis {
  float start;   // in_(0)
  float scale;   // per-sample step
  float limit;   // dim - 1 (in sampled-map coordinates)
};

// TF crop_and_resize_op.cc: in = lo_n * (dim-1) + i * scale, scale = (hi_n - lo_n)*(dim-1)/(crop-1)
__device__ __forceinline__ Axis make_axis(float lo_n, float hi_n, int dim, int crop, float* single) {
  Axis a;
  a.limit = (float)(dim - 1);
  a.scale = (crop > 1) ? (hi_n - lo_n) * a.limit / (float)(crop - 1) : 0.0f;
  a.start = lo_n * a.limit;
  *single = 0.5f * (lo_n + hi_n) * a.limit;   // crop == 1 path
  return a;
}

template <int POOL>
__global__ void __launch_bounds__(256) k_roi_pool(RoiParams p, const float4* __restrict__ rois,
                                                  const int32_t* __restrict__ roi_level,
                                                  const int32_t* __restrict__ count_dev, float* __restrict__ out) {
  // XCD-aware remap: hardware deals workgroups round-robin over the 8 XCDs
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int lb = xcd * p.blocks_per_xcd + slot;
  if (slot >= p.blocks_per_xcd || lb >= p.nblocks) return;
  const int lane = threadIdx.x & 63;
  const int task = lb * 4 + (threadIdx.x >> 6);
  const int PP = p.P * p.P;
  if (task >= p.n * PP) return;
  const int r = task / PP;
  const int bin = task - r * PP;
  const int py = bin / p.P, px = bin - py * p.P;
  const int C = p.C;
  float* obase = out + ((size_t)task) * C;

  const int cnt = count_dev ? min(*count_dev, p.n) : p.n;
  if (r >= cnt) {
    for (int c = lane * 4; c < C; c += 256) *reinterpret_cast<float4*>(obase + c) = make_float4(0, 0, 0, 0);
    return;
  }

  const int lvl = roi_level ? min(max(roi_level[r], 0), p.num_levels - 1) : 0;
  const float* __restrict__ feat = p.data[lvl];
  const int H = p.H[lvl], W = p.W[lvl];
  const float4 roi = rois[r];
  const int S = (POOL == ODET_ROI_POOL_NONE) ? 1 : 2;
  const int crop = p.P * S;

  // normalised box (y1,x1,y2,x2) exactly as the reference builds it
  float y1n, x1n, y2n, x2n;
  int Hs = H, Ws = W;     // dims of the map crop_and_resize samples (padded for TP_ALIGN)
  if (p.norm_mode == ODET_ROI_NORM_IMAGE) {
    y1n = roi.y / p.image_h; x1n = roi.x / p.image_w;            // roi_pooling.py:30-35
    y2n = roi.w / p.image_h; x2n = roi.z / p.image_w;
  } else if (p.norm_mode == ODET_ROI_NORM_STRIDE) {
    const float st = p.stride[lvl];
    const float hm = (float)(H - 1), wm = (float)(W - 1);
    y1n = (roi.y / st) / hm; x1n = (roi.x / st) / wm;            // roi_pooling.py:64,69-74
    y2n = (roi.w / st) / hm; x2n = (roi.z / st) / wm;
  } else {
    const float st = p.stride[lvl];
    const bool padded = (p.norm_mode == ODET_ROI_NORM_TP_ALIGN);
    const float off = padded ? 1.0f : 0.0f;
    if (padded) { Hs = H + 2; Ws = W + 2; }                      // roi_pooling.py:100
    float x0 = roi.x / st, y0 = roi.y / st;                      // :175
    float x1 = roi.z / st, y1 = roi.w / st;
    if (padded) { x0 = x0 + off; y0 = y0 + off; x1 = x1 + off; y1 = y1 + off; }   // :101
    const float cs = (float)crop;
    const float sw = (x1 - x0) / cs, sh = (y1 - y0) / cs;        // :120-121
    const float imh = (float)(Hs - 1), imw = (float)(Ws - 1);
    x1n = (x0 + sw / 2.0f - 0.5f) / imw;                         // :124
    y1n = (y0 + sh / 2.0f - 0.5f) / imh;                         // :125
    const float nw = sw * (float)(crop - 1) / imw;               // :127
    const float nh = sh * (float)(crop - 1) / imh;               // :128
    y2n = y1n + nh; x2n = x1n + nw;                              // :130
  }
  float ysingle, xsingle;
  const Axis ay = make_axis(y1n, y2n, Hs, crop, &ysingle);
  const Axis ax = make_axis(x1n, x2n, Ws, crop, &xsingle);
  const bool pad = (p.norm_mode == ODET_ROI_NORM_TP_ALIGN);

  // sample coordinates of this bin's S x S samples (wave-uniform)
  bool yok[2], xok[2];
  int ytop[2], ybot[2], xl[2], xr[2];
  float ylerp[2], xlerp[2];
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const int yy = py * S + s, xx = px * S + s;
    const float in_y = (crop > 1) ? ay.start + (float)yy * ay.scale : ysingle;
    const float in_x = (crop > 1) ? ax.start + (float)xx * ax.scale : xsingle;
    // TF: extrapolate when (in < 0 || in > dim-1).  Written as the positive test so that a NaN
    // coordinate can never turn into a tap index.
    yok[s] = (in_y >= 0.0f && in_y <= ay.limit);
    xok[s] = (in_x >= 0.0f && in_x <= ax.limit);
    const float fy = floorf(in_y), fx = floorf(in_x);
    ylerp[s] = in_y - fy;
    xlerp[s] = in_x - fx;
    int t = (int)fy, bt = (int)ceilf(in_y), l = (int)fx, rr = (int)ceilf(in_x);
    if (pad) {   // SYMMETRIC 1-px pad == edge replicate: padded[i] = src[clamp(i-1)]
      t = min(max(t - 1, 0), H - 1); bt = min(max(bt - 1, 0), H - 1);
      l = min(max(l - 1, 0), W - 1); rr = min(max(rr - 1, 0), W - 1);
    }
    ytop[s] = t; ybot[s] = bt; xl[s] = l; xr[s] = rr;
  }

  for (int c = lane * 4; c < C; c += 256) {
    float4 v[2][2];
#pragma unroll
    for (int sy = 0; sy < S; ++sy) {
#pragma unroll
      for (int sx = 0; sx < S; ++sx) {
        float4 res = make_float4(0, 0, 0, 0);
        if (yok[sy] && xok[sx]) {
          const float* rt = feat + ((size_t)ytop[sy] * W) * C + c;
          const float* rb = feat + ((size_t)ybot[sy] * W) * C + c;
          const float4 tl = *reinterpret_cast<const float4*>(rt + (size_t)xl[sx] * C);
          const float4 tr = *reinterpret_cast<const float4*>(rt + (size_t)xr[sx] * C);
          const float4 bl = *reinterpret_cast<const float4*>(rb + (size_t)xl[sx] * C);
          const float4 br = *reinterpret_cast<const float4*>(rb + (size_t)xr[sx] * C);
          const float xw = xlerp[sx], yw = ylerp[sy];
          float t, b;
          t = tl.x + (tr.x - tl.x) * xw; b = bl.x + (br.x - bl.x) * xw; res.x = t + (b - t) * yw;
          t = tl.y + (tr.y - tl.y) * xw; b = bl.y + (br.y - bl.y) * xw; res.y = t + (b - t) * yw;
          t = tl.z + (tr.z - tl.z) * xw; b = bl.z + (br.z - bl.z) * xw; res.z = t + (b - t) * yw;
          t = tl.w + (tr.w - tl.w) * xw; b = bl.w + (br.w - bl.w) * xw; res.w = t + (b - t) * yw;
        }
        v[sy][sx] = res;
      }
    }
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
    *reinterpret_cast<float4*>(obase + c) = o;
  }
}

extern "C" int odet_roi_pool(const odet_level_t* levels, int num_levels, int C, const float* rois,
                             const int32_t* roi_level, int n, const int32_t* count_dev, int norm_mode,
                             int image_h, int image_w, int pool_size, int pool_mode, float* out,
                             odet_stream_t stream) {
  ODET_REQUIRE(n >= 0, "odet_roi_pool: negative n");
  if (n == 0) return ODET_OK;
  ODET_REQUIRE(levels && rois && out, "odet_roi_pool: null pointer");
  ODET_REQUIRE(num_levels > 0 && num_levels <= ODET_MAX_LEVELS, "odet_roi_pool: num_levels %d out of range", num_levels);
  ODET_REQUIRE(C > 0 && (C & 3) == 0, "odet_roi_pool: C must be a positive multiple of 4 (got %d)", C);
  ODET_REQUIRE(pool_size > 0 && pool_size <= 64, "odet_roi_pool: pool_size out of range");
  ODET_REQUIRE(norm_mode >= 0 && norm_mode <= 3, "odet_roi_pool: unknown norm_mode %d", norm_mode);
  ODET_REQUIRE(pool_mode >= 0 && pool_mode <= 2, "odet_roi_pool: unknown pool_mode %d", pool_mode);
  ODET_REQUIRE(num_levels == 1 || roi_level, "odet_roi_pool: roi_level required with several levels");
  if (norm_mode == ODET_ROI_NORM_IMAGE) ODET_REQUIRE(image_h > 0 && image_w > 0, "odet_roi_pool: bad image shape");
  RoiParams p;
  for (int l = 0; l < ODET_MAX_LEVELS; ++l) {
    const odet_level_t* L = &levels[l < num_levels ? l : 0];
    ODET_REQUIRE(L->data && L->H > 0 && L->W > 0, "odet_roi_pool: bad level %d", l);
    if (norm_mode != ODET_ROI_NORM_IMAGE) ODET_REQUIRE(L->stride > 0.0f, "odet_roi_pool: bad stride on level %d", l);
    p.data[l] = L->data; p.H[l] = L->H; p.W[l] = L->W; p.stride[l] = L->stride;
  }
  p.num_levels = num_levels;
  p.C = C; p.n = n; p.norm_mode = norm_mode; p.P = pool_size; p.pool_mode = pool_mode;
  p.image_h = (float)image_h; p.image_w = (float)image_w;
  int64_t tasks = (int64_t)n * pool_size * pool_size;
  ODET_REQUIRE(tasks < (1ll << 30), "odet_roi_pool: too many output bins");
  p.nblocks = (int)((tasks + 3) / 4);
  p.blocks_per_xcd = (p.nblocks + 7) / 8;
  dim3 grid(p.blocks_per_xcd * 8), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (pool_mode == ODET_ROI_POOL_NONE)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_roi_pool<ODET_ROI_POOL_NONE>), grid, block, 0, st, p, (const float4*)rois,
                       roi_level, count_dev, out);
  else if (pool_mode == ODET_ROI_POOL_MAX2)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_roi_pool<ODET_ROI_POOL_MAX2>), grid, block, 0, st, p, (const float4*)rois,
                       roi_level, count_dev, out);
  else
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_roi_pool<ODET_ROI_POOL_AVG2>), grid, block, 0, st, p, (const float4*)rois,
                       roi_level, count_dev, out);
  ODET_LAUNCH_CHECK();
  return ODET_OK;
}

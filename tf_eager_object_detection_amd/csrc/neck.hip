// FPN neck, top-down merge: P_k = 0.5 * resize_bilinear(P_{k+1}, size(C_k)) + 0.5 * lateral(C_k)
// (reference model/fpn/resnet_fpn.py:385-398: tf.image.resize_bilinear of TF 1.x, align_corners=False,
// then keras Add of the two halves).  SURVEY 8(f) rank 3 -- the one part of the neck that is not a dense
// contraction: HBM-bound (lateral in, P_k out; the coarse map is 1/4 of the size and stays in L2), NHWC so a
// pixel's C channels are one coalesced segment.  Same float32 operation order as the TF kernel
// (resize_bilinear_op.cc): scale = in / out; src = dst * scale; lo = floor; hi = min(lo + 1, in - 1);
// top = tl + (tr - tl) * xl; bot = bl + (br - bl) * xl; up = top + (bot - top) * yl; out = up*0.5 + lat*0.5.
#include <hip/hip_fp16.h>

#include "odet_internal.h"

typedef uint32_t neck_u4 __attribute__((ext_vector_type(4)));   // 16 B, the type the non-temporal builtins take

template <typename FT>
struct NeckVec;
template <>
struct NeckVec<float> {                 // 4 channels = 16 B per lane
  static constexpr int N = 4;
  typedef float4 raw;
  static __device__ __forceinline__ void unpack(const raw& r, float (&f)[8]) { f[0] = r.x; f[1] = r.y; f[2] = r.z; f[3] = r.w; }
  static __device__ __forceinline__ raw pack(const float (&f)[8]) { return make_float4(f[0], f[1], f[2], f[3]); }
};
template <>
struct NeckVec<__half> {                // 8 channels = 16 B per lane
  static constexpr int N = 8;
  typedef uint4 raw;
  static __device__ __forceinline__ void unpack(const raw& r, float (&f)[8]) {
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float2 t = __half22float2(*reinterpret_cast<const __half2*>(&w[i]));
      f[2 * i] = t.x; f[2 * i + 1] = t.y;
    }
  }
  static __device__ __forceinline__ raw pack(const float (&f)[8]) {
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const __half2 t = __floats2half2_rn(f[2 * i], f[2 * i + 1]);
      w[i] = *reinterpret_cast<const uint32_t*>(&t);
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
  }
};

struct NeckParams {
  const void* top; const void* lat; void* out;
  int h, w, H, W, C;            // coarse map h x w, fine map H x W
  int vec_per_px;               // C / N
  long long total;              // B * H * W * vec_per_px
  float ys, xs;                 // h / H, w / W (float32, as TF computes them)
};

template <typename FT>
__global__ void __launch_bounds__(256) k_fpn_topdown_merge(NeckParams p) {
  typedef NeckVec<FT> V;
  typedef typename V::raw raw;
  const raw* __restrict__ top = reinterpret_cast<const raw*>(p.top);
  const raw* __restrict__ lat = reinterpret_cast<const raw*>(p.lat);
  raw* __restrict__ out = reinterpret_cast<raw*>(p.out);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < p.total; i += (long long)gridDim.x * 256) {
    const int v = (int)(i % p.vec_per_px);
    long long px = i / p.vec_per_px;
    const int x = (int)(px % p.W);
    px /= p.W;
    const int y = (int)(px % p.H);
    const int b = (int)(px / p.H);
    const float fy = (float)y * p.ys, fx = (float)x * p.xs;
    const float y0f = floorf(fy), x0f = floorf(fx);
    const int y0 = (int)y0f, x0 = (int)x0f;
    const int y1 = min(y0 + 1, p.h - 1), x1 = min(x0 + 1, p.w - 1);
    const float yl = fy - y0f, xl = fx - x0f;
    const long long tb = (long long)b * p.h * p.w;
    float tl[8], tr[8], bl[8], br[8], la[8], o[8];
    V::unpack(top[(tb + (long long)y0 * p.w + x0) * p.vec_per_px + v], tl);
    V::unpack(top[(tb + (long long)y0 * p.w + x1) * p.vec_per_px + v], tr);
    V::unpack(top[(tb + (long long)y1 * p.w + x0) * p.vec_per_px + v], bl);
    V::unpack(top[(tb + (long long)y1 * p.w + x1) * p.vec_per_px + v], br);
    {                                                        // the lateral map is streamed once
      const neck_u4 t = __builtin_nontemporal_load(reinterpret_cast<const neck_u4*>(&lat[i]));
      raw r;
      __builtin_memcpy(&r, &t, 16);
      V::unpack(r, la);
    }
#pragma unroll
    for (int k = 0; k < V::N; ++k) {
      const float t = tl[k] + (tr[k] - tl[k]) * xl;
      const float bt = bl[k] + (br[k] - bl[k]) * xl;
      const float up = t + (bt - t) * yl;
      o[k] = up * 0.5f + la[k] * 0.5f;
    }
    out[i] = V::pack(o);
  }
}

extern "C" int odet_fpn_topdown_merge(const void* top, int h, int w, const void* lateral, int H, int W, int B,
                                      int C, void* out, int f16, odet_stream_t stream) {
  ODET_REQUIRE(top && lateral && out, "odet_fpn_topdown_merge: null pointer");
  ODET_REQUIRE(h > 0 && w > 0 && H > 0 && W > 0 && B >= 0 && C > 0, "odet_fpn_topdown_merge: bad sizes");
  const int n = f16 ? 8 : 4;
  ODET_REQUIRE(C % n == 0, "odet_fpn_topdown_merge: C must be a multiple of %d (got %d)", n, C);
  if (B == 0) return ODET_OK;
  NeckParams p;
  p.top = top; p.lat = lateral; p.out = out;
  p.h = h; p.w = w; p.H = H; p.W = W; p.C = C;
  p.vec_per_px = C / n;
  p.total = (long long)B * H * W * p.vec_per_px;
  p.ys = (float)h / (float)H;
  p.xs = (float)w / (float)W;
  const long long blocks = (p.total + 255) / 256;
  const int grid = (int)std::min<long long>(blocks, 256 * 64);      // grid-stride beyond 64 workgroups per CU
  if (f16)
    hipLaunchKernelGGL(k_fpn_topdown_merge<__half>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
  else
    hipLaunchKernelGGL(k_fpn_topdown_merge<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
  ODET_LAUNCH_CHECK();
  return ODET_OK;
}

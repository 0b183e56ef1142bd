// The ResNet stem in ONE launch (model/fpn/resnet_fpn.py:262-289, resnet_faster_rcnn.py:31-60):
//     ZeroPadding2D(3) -> Conv2D(64, 7x7, stride 2, 'valid') -> frozen BN (folded) -> ReLU -> ZeroPadding2D(1) ->
//     MaxPooling2D(3x3, stride 2, 'valid')
// from the NHWC image (float32 or float16, 3 channels) to the NHWC float16 pooled map [B, PH, PW, 64].  As library
// convolution + pooling pass the 64-channel convolution output (273 MB at batch 8, 800x1333) is written and read back
// and the convolution needs a zero-fill + split-K reduction (0.40 ms together); here neither the padded image nor
// the convolution output ever exist in memory: 51 MB in, 68 MB out.
//
//  * WORKGROUP (4 waves) = a tile of 8 x 8 POOLED pixels.  It needs the 17 x 17 convolution pixels around them and,
//    for those, a 39 x 39 patch of the image.  The patch is staged in LDS as [39][40] pixels x 4 channels of float16 (the
//    fourth channel is zero: 8 bytes per pixel, so that every fragment below is one aligned 16-byte read); pixels
//    outside the image are the zero padding.
//  * The convolution is an implicit GEMM on the matrix cores, K = 7 kernel rows x 32 (7 taps x 4 channels = 28 used,
//    the rest multiplied by zero weights): v_mfma_f32_16x16x32_f16 computes D = W_tile . X_tile^T with the MFMA's A
//    operand = 16 output channels (the weights of a kernel row, 16 bytes per lane, held in registers for the whole
//    workgroup: 4 channel tiles x 7 rows) and its B operand = 16 convolution pixels: lane (pixel, q) reads the 16
//    bytes of patch pixels 2 cx + 2 q, 2 cx + 2 q + 1 of row 2 cy + ky -- consecutive pixels are 16 bytes apart:
//    conflict-free.  A lane ends up with 4 consecutive channels of one pixel: + bias, ReLU, one rounding, 8 bytes into
//    the LDS image of the convolution tile [17 x 17 pixels][64 channels] (pixels outside the convolution's output are
//    written as zero = the pooling's padding).
//  * After a barrier every thread takes 16 channels of one pooled pixel: the maximum over its 3 x 3 convolution
//    pixels (32-byte LDS reads), 32 contiguous bytes out.
//  * 51 KB of LDS and 256 threads per workgroup: three workgroups per CU, whose staging / MFMA / pooling phases overlap
//    (a 8 x 16 tile with 512 threads, one workgroup per CU: 237 us at batch 8 instead of the figure below).
#include <hip/hip_fp16.h>

#include <mutex>

#include "odet_internal.h"

typedef _Float16 st_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 st_h4 __attribute__((ext_vector_type(4)));
typedef float st_f4 __attribute__((ext_vector_type(4)));

#define ST_PH 8                       // pooled rows of a tile
#define ST_PW 8                       // pooled columns
#define ST_CH (2 * ST_PH + 1)         // 17 convolution rows
#define ST_CW (2 * ST_PW + 1)         // 17 convolution columns
#define ST_IH (2 * ST_CH + 5)         // 39 image rows
#define ST_IW (2 * ST_CW + 6)         // 39 image columns + 1 (the fragment of the last column reads one pixel further)
#define ST_THREADS 256                // 4 waves: three workgroups (51 KB of LDS each) share a CU and overlap their phases
#define ST_PATCH_BYTES (ST_IH * ST_IW * 8)            // 12 480
#define ST_CONV_PIX (ST_CH * ST_CW)                   // 289
#define ST_CONV_TILES ((ST_CONV_PIX + 15) / 16)       // 19
#define ST_CONV_BYTES (ST_CONV_TILES * 16 * 128)      // 38 912
#define ST_LDS_BYTES (ST_PATCH_BYTES + ST_CONV_BYTES)

typedef unsigned int st_u4 __attribute__((ext_vector_type(4)));
typedef unsigned int st_u2 __attribute__((ext_vector_type(2)));
// m = max(m, [a | b]) per float16 element
__device__ __forceinline__ void st_pk_max(st_h8& m, const st_h4& a, const st_h4& b) {
  st_u4 mm = __builtin_bit_cast(st_u4, m);
  const st_u2 ua = __builtin_bit_cast(st_u2, a), ub = __builtin_bit_cast(st_u2, b);
  asm("v_pk_max_f16 %0, %0, %1" : "+v"(mm[0]) : "v"(ua[0]));
  asm("v_pk_max_f16 %0, %0, %1" : "+v"(mm[1]) : "v"(ua[1]));
  asm("v_pk_max_f16 %0, %0, %1" : "+v"(mm[2]) : "v"(ub[0]));
  asm("v_pk_max_f16 %0, %0, %1" : "+v"(mm[3]) : "v"(ub[1]));
  m = __builtin_bit_cast(st_h8, mm);
}

struct StemParams {
  const void* img; int img_f16;       // [B][H][W][3] float32 (0) or float16 (1)
  const _Float16* w;                  // packed in fragment order [4 channel tiles][7 kernel rows][64 lanes][8]: lane (q, channel)
                                      // holds k = 8 q .. 8 q + 7 of (channel, kernel row), k = 4 * kx + c, zero where kx == 7 or c == 3
  const _Float16* bias;               // [64]
  _Float16* out;                      // [B][PH][PW][64]
  int B, H, W, CH, CW, PH, PW;        // image, convolution output and pooled sizes
  int tiles_x, tiles_y;
};

__global__ void __launch_bounds__(ST_THREADS) k_stem_conv7_pool3(StemParams p) {
  extern __shared__ __align__(16) unsigned char lds[];
  unsigned char* patch = lds;
  unsigned char* conv = lds + ST_PATCH_BYTES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, lq = lane >> 4;
  int blk = blockIdx.x;
  const int tx = blk % p.tiles_x; blk /= p.tiles_x;
  const int ty = blk % p.tiles_y;
  const int b = blk / p.tiles_y;
  const int py0 = ty * ST_PH, px0 = tx * ST_PW;        // first pooled pixel of the tile
  const int cy0 = 2 * py0 - 1, cx0 = 2 * px0 - 1;      // first convolution pixel (may be -1: pooling pad)
  const int iy0 = 2 * cy0 - 3, ix0 = 2 * cx0 - 3;      // first image pixel (may be negative: convolution pad)

  // ---- the weights of this wave's MFMAs: 4 channel tiles x 7 kernel rows, lane (channel l15, q): 16 bytes
  st_h8 wr[4][7];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int ky = 0; ky < 7; ++ky)
      wr[ct][ky] = *reinterpret_cast<const st_h8*>(p.w + ((ct * 7 + ky) * 64 + lane) * 8);      // 1 KB per wave instruction
  float bv[4][4];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int j = 0; j < 4; ++j) bv[ct][j] = (float)p.bias[ct * 16 + 4 * lq + j];

  // ---- stage the image patch: one pixel (3 values -> 4 halfs, 8 bytes) per thread and trip; the loads of all six
  // trips are issued before the first conversion (pixels outside the image are the zero padding)
  {
    constexpr int TRIPS = (ST_IH * ST_IW + ST_THREADS - 1) / ST_THREADS;        // 7
    const long long img_base = (long long)b * p.H * p.W * 3;
    float v[TRIPS][3];
    long long off[TRIPS];
    bool ok[TRIPS];
#pragma unroll
    for (int k = 0; k < TRIPS; ++k) {
      const int i = tid + ST_THREADS * k;
      const int r = i / ST_IW, c = i - r * ST_IW;
      const int y = iy0 + r, x = ix0 + c;
      ok[k] = i < ST_IH * ST_IW && y >= 0 && y < p.H && x >= 0 && x < p.W;
      off[k] = ok[k] ? img_base + ((long long)y * p.W + x) * 3 : 0;     // (a valid address either way)
    }
    // (the loads are UNCONDITIONAL -- the address is valid either way -- and the padding is selected afterwards: behind
    // `ok ? s[0] : 0` every trip was a branch of its own around three loads)
    if (p.img_f16) {
#pragma unroll
      for (int k = 0; k < TRIPS; ++k) {
        const _Float16* s = reinterpret_cast<const _Float16*>(p.img) + off[k];
        const _Float16 a0 = s[0], a1 = s[1], a2 = s[2];
        v[k][0] = (float)a0; v[k][1] = (float)a1; v[k][2] = (float)a2;
      }
    } else {
#pragma unroll
      for (int k = 0; k < TRIPS; ++k) {
        const float* s = reinterpret_cast<const float*>(p.img) + off[k];
        const float a0 = s[0], a1 = s[1], a2 = s[2];
        v[k][0] = a0; v[k][1] = a1; v[k][2] = a2;
      }
    }
#pragma unroll
    for (int k = 0; k < TRIPS; ++k)
      if (!ok[k]) { v[k][0] = 0.0f; v[k][1] = 0.0f; v[k][2] = 0.0f; }
#pragma unroll
    for (int k = 0; k < TRIPS; ++k) {
      const int i = tid + ST_THREADS * k;
      if (i < ST_IH * ST_IW) {
        const st_u2 hu = {d_cvt_pk_f16(v[k][0], v[k][1]), d_cvt_pk_f16(v[k][2], 0.0f)};
        const st_h4 h = __builtin_bit_cast(st_h4, hu);
        *reinterpret_cast<st_h4*>(patch + i * 8) = h;
      }
    }
  }
  __syncthreads();

  // ---- convolution: pixel tiles wv, wv + 4, ... of the 17 x 17 convolution pixels (raster order)
  for (int t = wv; t < ST_CONV_TILES; t += ST_THREADS / 64) {
    int cp = t * 16 + l15;                              // this lane's pixel as the MFMA's B column
    if (cp >= ST_CONV_PIX) cp = ST_CONV_PIX - 1;        // (the last tile's spare columns recompute the last pixel)
    const int cyl = cp / ST_CW, cxl = cp - cyl * ST_CW;
    const unsigned char* src = patch + ((2 * cyl) * ST_IW + 2 * cxl + 2 * lq) * 8;
    st_f4 acc[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) acc[ct] = (st_f4){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int ky = 0; ky < 7; ++ky) {
      const st_h8 xf = *reinterpret_cast<const st_h8*>(src + ky * ST_IW * 8);
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wr[ct][ky], xf, acc[ct], 0, 0, 0);
    }
    // D[row = channel 4 lq + j of tile ct][col = pixel l15]: + bias, ReLU; pixels outside the convolution output = 0
    const int op = t * 16 + l15;
    if (op < ST_CONV_PIX) {
      const int oy = op / ST_CW, ox = op - oy * ST_CW;
      const int cy = cy0 + oy, cx = cx0 + ox;
      const bool inside = cy >= 0 && cy < p.CH && cx >= 0 && cx < p.CW;
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        float vf[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float v = acc[ct][j] + bv[ct][j];
          vf[j] = (v < 0.0f || !inside) ? 0.0f : v;
        }
        const st_u2 ou = {d_cvt_pk_f16(vf[0], vf[1]), d_cvt_pk_f16(vf[2], vf[3])};
        const st_h4 o = __builtin_bit_cast(st_h4, ou);
        // (8-byte chunk ct * 4 + lq of the pixel's 128 bytes, XOR-swizzled by the pixel: the 16 pixels of a tile would
        // otherwise all write the same LDS bank -- 16-way conflicts that made this the longest phase of the kernel)
        *reinterpret_cast<st_h4*>(conv + op * 128 + (((ct * 4 + lq) ^ (op & 15)) * 8)) = o;
      }
    }
  }
  __syncthreads();

  // ---- 3 x 3 / 2 max-pooling: thread = (pooled pixel tid >> 2, channels 16 (tid & 3) .. + 15)
  {
    const int pp = tid >> 2, cg = (tid & 3) * 16;
    const int pyl = pp / ST_PW, pxl = pp - pyl * ST_PW;
    const int py = py0 + pyl, px = px0 + pxl;
    if (py < p.PH && px < p.PW) {
      st_h8 m0, m1;
#pragma unroll
      for (int e = 0; e < 8; ++e) { m0[e] = (_Float16)0.0f; m1[e] = (_Float16)0.0f; }   // (ReLU outputs and zero padding: >= 0)
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const int P = (2 * pyl + dy) * ST_CW + 2 * pxl + dx;
          const unsigned char* s = conv + P * 128;
          const int c0 = (tid & 3) * 4, sw = P & 15;
          const st_h4 q0 = *reinterpret_cast<const st_h4*>(s + ((c0 ^ sw) * 8));
          const st_h4 q1 = *reinterpret_cast<const st_h4*>(s + (((c0 + 1) ^ sw) * 8));
          const st_h4 q2 = *reinterpret_cast<const st_h4*>(s + (((c0 + 2) ^ sw) * 8));
          const st_h4 q3 = *reinterpret_cast<const st_h4*>(s + (((c0 + 3) ^ sw) * 8));
          // (packed maxima: the values are ReLU outputs or the zero padding, so v_pk_max_f16 == the comparison; written as
          // the instruction itself: the generic maximum canonicalises both operands first, twice the instructions)
          st_pk_max(m0, q0, q1);
          st_pk_max(m1, q2, q3);
        }
      _Float16* dst = p.out + (((long long)b * p.PH + py) * p.PW + px) * 64 + cg;
      *reinterpret_cast<st_h8*>(dst) = m0;
      *reinterpret_cast<st_h8*>(dst + 8) = m1;
    }
  }
}

// repack [64][3][7][7] (the framework's layout) / any strides -> the kernel's fragment order, zeros in the padding slots
__global__ void __launch_bounds__(256) k_stem_pack_weights(const _Float16* __restrict__ w, long long s_o, long long s_c,
                                                           long long s_y, long long s_x, _Float16* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 64 * 7 * 32) return;
  // fragment order: [channel tile ct][kernel row ky][lane = 16 q + channel-in-tile][8 halfs k = 8 q + e]
  const int e = i & 7, ln = (i >> 3) & 63, ky = (i >> 9) % 7, ct = i / (7 * 512);
  const int o = ct * 16 + (ln & 15), k = 8 * (ln >> 4) + e;
  const int kx = k >> 2, c = k & 3;
  _Float16 v = (_Float16)0.0f;
  if (kx < 7 && c < 3) v = w[o * s_o + c * s_c + ky * s_y + kx * s_x];
  out[i] = v;
}

extern "C" int odet_stem_pack_weights_f16(const void* w, long long stride_o, long long stride_c, long long stride_y,
                                          long long stride_x, void* packed, odet_stream_t stream) {
  ODET_REQUIRE(w && packed, "odet_stem_pack_weights_f16: null pointer");
  hipLaunchKernelGGL(k_stem_pack_weights, dim3((64 * 7 * 32 + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     (const _Float16*)w, stride_o, stride_c, stride_y, stride_x, (_Float16*)packed);
  ODET_LAUNCH_CHECK();
  return ODET_OK;
}

extern "C" int odet_stem_conv7_pool3_f16(const void* images, int images_f16, const void* packed_w, const void* bias, void* out,
                                         int batch, int H, int W, odet_stream_t stream) {
  ODET_REQUIRE(images && packed_w && bias && out, "odet_stem_conv7_pool3_f16: null pointer");
  ODET_REQUIRE(batch > 0 && H >= 7 && W >= 7, "odet_stem_conv7_pool3_f16: bad image shape");
  ODET_REQUIRE(((uintptr_t)packed_w | (uintptr_t)out) % 16 == 0, "odet_stem_conv7_pool3_f16: pointers must be 16-byte aligned");
  static std::once_flag once;
  static hipError_t once_rc = hipSuccess;
  std::call_once(once, [] {
    once_rc = hipFuncSetAttribute((const void*)k_stem_conv7_pool3, hipFuncAttributeMaxDynamicSharedMemorySize, ST_LDS_BYTES);
  });
  ODET_HIP(once_rc);
  StemParams p;
  p.img = images; p.img_f16 = images_f16 ? 1 : 0; p.w = (const _Float16*)packed_w; p.bias = (const _Float16*)bias;
  p.out = (_Float16*)out;
  p.B = batch; p.H = H; p.W = W;
  p.CH = (H + 6 - 7) / 2 + 1; p.CW = (W + 6 - 7) / 2 + 1;          // pad 3, 7x7, stride 2, 'valid'
  p.PH = (p.CH + 2 - 3) / 2 + 1; p.PW = (p.CW + 2 - 3) / 2 + 1;    // pad 1, 3x3, stride 2, 'valid'
  p.tiles_x = (p.PW + ST_PW - 1) / ST_PW; p.tiles_y = (p.PH + ST_PH - 1) / ST_PH;
  const long long blocks = (long long)p.tiles_x * p.tiles_y * batch;
  ODET_REQUIRE(blocks < (1ll << 31), "odet_stem_conv7_pool3_f16: too many workgroups");
  hipLaunchKernelGGL(k_stem_conv7_pool3, dim3((unsigned)blocks), dim3(ST_THREADS), ST_LDS_BYTES, (hipStream_t)stream, p);
  ODET_LAUNCH_CHECK();
  return ODET_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// The first convolution of VGG16 (vgg16_faster_rcnn.py:260-342: Conv2D(64, 3x3, padding 'same') + ReLU on the 3-channel
// image) in ONE launch from the NHWC image (float32 or float16) to the NHWC float16 map [B, H, W, 64].  As library
// convolution + epilogue pass the 64-channel output was written, read back and written again (1.5 ms for 32 images of
// 600 x 800 + the image's float16 copy); here it is written once: the kernel runs at the store rate of its output.
//
//  * WORKGROUP (4 waves) = 16 rows x 32 columns of output pixels.  The 19 x 36 image pixels around them are staged in LDS
//    as 4 halfs per pixel (the fourth is zero; pixels outside the image are the zero padding).
//  * Implicit GEMM on the matrix cores, the stem kernel's way: D = W_tile . X_tile^T with v_mfma_f32_16x16x32_f16, the B
//    operand = 16 consecutive pixels of a row.  One MFMA takes TWO kernel rows: lane (pixel x, q) reads the 16 bytes of
//    patch pixels x - 1 + 2 (q & 1), x + 2 (q & 1) of kernel row 2 kp + (q >> 1) -- k = 8 q + 4 (pixel) + channel, so
//    K = 32 holds 2 rows x 4 pixels x 4 channels (27 of the 64 products of the two MFMAs are real, the rest meet zero
//    weights; the matrix pipe is idle most of the time either way).
//  * The weight rows are packed so that a lane's results of channel tiles (0, 1) / (2, 3) are 8 consecutive channels:
//    with the four lanes of a pixel one store instruction writes 64 contiguous bytes, a pixel tile 2 KB.
#define RG_TH 16                      // output rows of a tile (8: 650 us, 16: 618 us, 32: 626 us for 32 images of 600 x 800)
#define RG_TW 32                      // output columns
#define RG_PR (RG_TH + 3)             // patch rows: y0 - 1 .. y0 + RG_TH + 1 (kernel row "3" of the second MFMA meets zero weights)
#define RG_PC (RG_TW + 4)             // patch columns: x0 - 1 .. x0 + 34
#define RG_THREADS 256

struct RgbConvParams {
  const void* img; int img_f16;       // [B][H][W][3]
  const _Float16* w;                  // packed [4 channel tiles][2 row pairs][64 lanes][8] (odet_conv3x3_rgb_pack_weights_f16)
  const _Float16* bias;               // [64]
  _Float16* out;                      // [B][H][W][64]
  int B, H, W, relu;
  int tiles_x, tiles_y;
};

__global__ void __launch_bounds__(RG_THREADS) k_conv3x3_rgb_f16(RgbConvParams p) {
  __shared__ __align__(16) unsigned char patch[RG_PR * RG_PC * 8];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, lq = lane >> 4;
  int blk = blockIdx.x;
  const int tx = blk % p.tiles_x; blk /= p.tiles_x;
  const int ty = blk % p.tiles_y;
  const int b = blk / p.tiles_y;
  const int y0 = ty * RG_TH, x0 = tx * RG_TW;

  st_h8 wr[4][2];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int kp = 0; kp < 2; ++kp) wr[ct][kp] = *reinterpret_cast<const st_h8*>(p.w + ((ct * 2 + kp) * 64 + lane) * 8);
  // this lane's channels: 8 lq .. + 7 (tiles 0, 1) and 32 + 8 lq .. + 7 (tiles 2, 3)
  const st_h8 b0 = *reinterpret_cast<const st_h8*>(p.bias + 8 * lq), b1 = *reinterpret_cast<const st_h8*>(p.bias + 32 + 8 * lq);

  {
    constexpr int TRIPS = (RG_PR * RG_PC + RG_THREADS - 1) / RG_THREADS;        // 3
    const long long img_base = (long long)b * p.H * p.W * 3;
    float v[TRIPS][3];
#pragma unroll
    for (int k = 0; k < TRIPS; ++k) {
      const int i = tid + RG_THREADS * k;
      const int r = i / RG_PC, c = i - r * RG_PC;
      const int y = y0 - 1 + r, x = x0 - 1 + c;
      const bool ok = i < RG_PR * RG_PC && y >= 0 && y < p.H && x >= 0 && x < p.W;
      const long long o = ok ? img_base + ((long long)y * p.W + x) * 3 : 0;
      if (p.img_f16) {
        const _Float16* s = reinterpret_cast<const _Float16*>(p.img) + o;
        v[k][0] = ok ? (float)s[0] : 0.0f; v[k][1] = ok ? (float)s[1] : 0.0f; v[k][2] = ok ? (float)s[2] : 0.0f;
      } else {
        const float* s = reinterpret_cast<const float*>(p.img) + o;
        v[k][0] = ok ? s[0] : 0.0f; v[k][1] = ok ? s[1] : 0.0f; v[k][2] = ok ? s[2] : 0.0f;
      }
    }
#pragma unroll
    for (int k = 0; k < TRIPS; ++k) {
      const int i = tid + RG_THREADS * k;
      if (i < RG_PR * RG_PC) {
        const st_u2 hu = {d_cvt_pk_f16(v[k][0], v[k][1]), d_cvt_pk_f16(v[k][2], 0.0f)};
        const st_h4 h = __builtin_bit_cast(st_h4, hu);
        *reinterpret_cast<st_h4*>(patch + i * 8) = h;
      }
    }
  }
  __syncthreads();

  // pixel tiles: 16 consecutive pixels of a row; tile t = (row t >> 1, columns 16 (t & 1) ..); a wave takes RG_TH / 2
#pragma unroll
  for (int n = 0; n < RG_TH / 2; ++n) {
    const int t = wv * (RG_TH / 2) + n;
    const int ry = t >> 1, cx = (t & 1) * 16 + l15;                 // output pixel inside the tile
    // patch pixel (row ry + 2 kp + (lq >> 1), column cx + 2 (lq & 1)) and its right neighbour: 16 bytes at an 8-byte
    // aligned address (two 8-byte reads)
    const unsigned char* src = patch + ((ry + (lq >> 1)) * RG_PC + cx + 2 * (lq & 1)) * 8;
    st_f4 acc[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) acc[ct] = (st_f4){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int kp = 0; kp < 2; ++kp) {
      const st_h4 lo = *reinterpret_cast<const st_h4*>(src + kp * 2 * RG_PC * 8);
      const st_h4 hi = *reinterpret_cast<const st_h4*>(src + kp * 2 * RG_PC * 8 + 8);
      const st_h8 xf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wr[ct][kp], xf, acc[ct], 0, 0, 0);
    }
    const int y = y0 + ry, x = x0 + cx;
    if (y < p.H && x < p.W) {
      float f0[8], f1[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float v0 = acc[e >> 2][e & 3] + (float)b0[e];
        float v1 = acc[2 + (e >> 2)][e & 3] + (float)b1[e];
        if (p.relu) { v0 = v0 < 0.0f ? 0.0f : v0; v1 = v1 < 0.0f ? 0.0f : v1; }
        f0[e] = v0; f1[e] = v1;
      }
      const st_h8 o0 = d_cvt8_f16<st_h8>(f0), o1 = d_cvt8_f16<st_h8>(f1);
      _Float16* dst = p.out + (((long long)b * p.H + y) * p.W + x) * 64 + 8 * lq;
      *reinterpret_cast<st_h8*>(dst) = o0;
      *reinterpret_cast<st_h8*>(dst + 32) = o1;
    }
  }
}

// repack [64][3][3][3] (the framework's layout, any strides) -> the kernel's fragment order:
// [channel tile ct][row pair kp][lane = 16 q + row i of the MFMA's A operand][8 halfs k = 8 q + e]; MFMA row i of tile ct is
// output channel 32 (ct >> 1) + 8 (i >> 2) + 4 (ct & 1) + (i & 3); k = 8 q + e is kernel row 2 kp + (q >> 1), kernel column
// 2 (q & 1) + (e >> 2), input channel e & 3 -- zero where the row or column is 3 or the channel is 3
__global__ void __launch_bounds__(256) k_conv3x3_rgb_pack_weights(const _Float16* __restrict__ w, long long s_o, long long s_c,
                                                                  long long s_y, long long s_x, _Float16* __restrict__ out) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= 4 * 2 * 64 * 8) return;
  const int e = idx & 7, ln = (idx >> 3) & 63, kp = (idx >> 9) & 1, ct = idx >> 10;
  const int i = ln & 15, q = ln >> 4;
  const int o = 32 * (ct >> 1) + 8 * (i >> 2) + 4 * (ct & 1) + (i & 3);
  const int ky = 2 * kp + (q >> 1), kx = 2 * (q & 1) + (e >> 2), c = e & 3;
  _Float16 v = (_Float16)0.0f;
  if (ky < 3 && kx < 3 && c < 3) v = w[o * s_o + c * s_c + ky * s_y + kx * s_x];
  out[idx] = v;
}

extern "C" int odet_conv3x3_rgb_pack_weights_f16(const void* w, long long stride_o, long long stride_c, long long stride_y,
                                                 long long stride_x, void* packed, odet_stream_t stream) {
  ODET_REQUIRE(w && packed, "odet_conv3x3_rgb_pack_weights_f16: null pointer");
  hipLaunchKernelGGL(k_conv3x3_rgb_pack_weights, dim3((4 * 2 * 64 * 8 + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     (const _Float16*)w, stride_o, stride_c, stride_y, stride_x, (_Float16*)packed);
  ODET_LAUNCH_CHECK();
  return ODET_OK;
}

extern "C" int odet_conv3x3_rgb_f16(const void* images, int images_f16, const void* packed_w, const void* bias, void* out,
                                    int batch, int H, int W, int relu, odet_stream_t stream) {
  ODET_REQUIRE(images && packed_w && bias && out, "odet_conv3x3_rgb_f16: null pointer");
  ODET_REQUIRE(batch > 0 && H > 0 && W > 0, "odet_conv3x3_rgb_f16: bad image shape");
  ODET_REQUIRE(((uintptr_t)packed_w | (uintptr_t)bias | (uintptr_t)out) % 16 == 0, "odet_conv3x3_rgb_f16: pointers must be 16-byte aligned");
  RgbConvParams p;
  p.img = images; p.img_f16 = images_f16 ? 1 : 0; p.w = (const _Float16*)packed_w; p.bias = (const _Float16*)bias;
  p.out = (_Float16*)out; p.B = batch; p.H = H; p.W = W; p.relu = relu ? 1 : 0;
  p.tiles_x = (W + RG_TW - 1) / RG_TW; p.tiles_y = (H + RG_TH - 1) / RG_TH;
  const long long blocks = (long long)p.tiles_x * p.tiles_y * batch;
  ODET_REQUIRE(blocks < (1ll << 31), "odet_conv3x3_rgb_f16: too many workgroups");
  hipLaunchKernelGGL(k_conv3x3_rgb_f16, dim3((unsigned)blocks), dim3(RG_THREADS), 0, (hipStream_t)stream, p);
  ODET_LAUNCH_CHECK();
  return ODET_OK;
}

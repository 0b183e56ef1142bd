// EXPERIMENT, not built into libodet_hip.so: the DIRECT-TO-REGISTER first versions of csrc/conv1x1.hip (every wave
// loads its weight fragments from global memory / L2 itself; no LDS).  tools/exp/conv1x1_mfma_build.sh +
// conv1x1_mfma_bench.py build and run it as odet_conv1x1_f16_direct next to the product kernel.
// Measured on MI355X (batch 8 at 800x1333, float16 NHWC), with shortcut; library = convolution + epilogue pass:
//                               library   direct v1   direct v2 (this file)   LDS-staged (product)
//     64 -> 256  at 200x334 :   224 us     237 us         219 us                  157 us
//     128 -> 512 at 100x167 :   110 us     174 us         151 us                   91 us
//     256 -> 1024 at 50x84  :    64 us     111 us         117 us                   57 us
// v1 loaded the shortcut after the MFMAs; v2 requests it a group ahead.  Without the shortcut v2 still takes 139 /
// 108 / 103 us: every wave re-reads its weight rows from L2 for each 32-pixel slab (550 MB of L2 traffic on the
// 50x84 layer) and feeds them straight into the MFMAs at 2-3 waves per SIMD -- which is what the workgroup tile
// with the weight group staged once in LDS (the product kernel) removes.
//
// 1x1 stride-1 convolution of the dense path with its whole epilogue, on the matrix cores (SURVEY 8(f) rank 3:
// "Backbone + FPN neck on MFMA ... frozen-BN folded into conv"): the third convolution of every bottleneck block
// (resnet_fpn.py:154-205) and what follows it,
//     y[m, :] = relu( x[m, :] . W^T + bias + shortcut[m, :] )          x [M, K], W [N, K], y / shortcut [M, N]
// in NHWC float16 with float32 accumulation and ONE rounding.  As "library convolution + epilogue pass" the
// [M, N] output makes three more trips through HBM (written by the convolution, read and re-written by the
// epilogue next to the shortcut); these layers are bound by that traffic, not by the contraction (K = 64 .. 256:
// 32 .. 128 FLOP per output byte), so the kernel is built around the streams, not around the MFMA rate:
//
//  * one WAVE owns a tile of 32 pixels x NT channels and v_mfma_f32_32x32x16_f16 computes it TRANSPOSED
//    (D = W_tile . x_tile^T): the MFMA's A operand is 32 rows of W, its B operand 32 pixels of x.  Both fragments
//    are "8 consecutive k of one row" = one 16-byte global load per lane straight into the operand registers,
//    no LDS and no shuffles; the x fragments of the whole K stay in registers across the channel loop, so x is
//    read from HBM exactly once.  W (<= 512 KB) is served by L1 / L2.
//  * in the transposed result a lane holds ONE pixel and 16 channels per 32x32 block.  The rows of W are fed in a
//    permuted order (row i of the MFMA = channel 32*((i>>2)&1) + 16*j + 4*(i>>3) + (i&3) of the 64-channel group,
//    j = block 0 / 1), which makes those 16 registers 16 CONSECUTIVE channels and the two blocks of a group 32
//    consecutive channels: the shortcut is read and the output written with 16-byte accesses, 64 contiguous bytes
//    per lane, a full 128-byte line per pixel from the two lane halves.
//
// HBM bytes per call: M*K*2 (x) + M*N*2 (shortcut) + M*N*2 (y) -- the algorithmic minimum.
#include <hip/hip_fp16.h>

#include "odet_internal.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

struct Conv1x1DirectParams {
  const _Float16* x; const _Float16* w; const _Float16* bias; const _Float16* res; _Float16* y;
  long long M;
  int K, N, relu;
  int tiles_n;      // N / NT
  int nt;           // channels per wave tile (multiple of 64)
  long long waves;  // ceil(M / 32) * tiles_n
};

__device__ __forceinline__ h8 ldg16(const _Float16* p) { return *reinterpret_cast<const h8*>(p); }

// one 64-channel group of a wave's tile: weights -> 2 x KSTEPS MFMAs -> bias + shortcut (already in `cur`) -> store;
// the NEXT group's shortcut is requested into `nxt` before the weights
template <int KSTEPS>
__device__ __forceinline__ void conv1x1_direct_group(const Conv1x1DirectParams& p, const h8 (&xa)[KSTEPS], int n0, int n_end, int perm,
                                              int h, long long mc, bool store, const h8 (&cur)[4], h8 (&nxt)[4]) {
  const int K = p.K, N = p.N;
  const _Float16* w0 = p.w + (long long)(n0 + perm) * K + 8 * h;
  const _Float16* w1 = w0 + 16 * K;
  const int c0 = n0 + 32 * h;
  const long long off = mc * N + c0;
  if (p.res && n0 + 64 < n_end) {
#pragma unroll
    for (int q = 0; q < 4; ++q) nxt[q] = ldg16(p.res + off + 64 + 8 * q);
  }
  f16v acc0, acc1;
#pragma unroll
  for (int i = 0; i < 16; ++i) { acc0[i] = 0.0f; acc1[i] = 0.0f; }
#pragma unroll
  for (int s = 0; s < KSTEPS; ++s) {
    const h8 a0 = ldg16(w0 + 16 * s);
    const h8 a1 = ldg16(w1 + 16 * s);
    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, xa[s], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, xa[s], acc1, 0, 0, 0);
  }
  // lane (pixel r, half h): channels n0 + 32 h + [0, 32): acc0 -> +0..15, acc1 -> +16..31
  const _Float16* bp = p.bias + c0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const h8 bv = ldg16(bp + 8 * q);
    h8 ov;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int idx = (q & 1) * 8 + e;
      float v = (q < 2 ? acc0[idx] : acc1[idx]) + (float)bv[e];
      if (p.res) v = v + (float)cur[q][e];
      if (p.relu) v = (v < 0.0f) ? 0.0f : v;
      ov[e] = (_Float16)v;
    }
    if (store) *reinterpret_cast<h8*>(p.y + off + 8 * q) = ov;
  }
}

// KSTEPS = K / 16 (4, 8 or 16): the pixel fragments of the whole K live in registers
template <int KSTEPS>
__global__ void __launch_bounds__(256) k_conv1x1_f16_direct(Conv1x1DirectParams p) {
  const int lane = threadIdx.x & 63;
  const long long gw = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (gw >= p.waves) return;                       // (whole waves leave; no barrier in this kernel)
  const long long slab = gw / p.tiles_n;
  const int tn = (int)(gw - slab * p.tiles_n);
  const int r = lane & 31, h = lane >> 5;
  const int K = p.K, N = p.N;
  const long long m = slab * 32 + r;
  const long long mc = m < p.M ? m : p.M - 1;      // rows past the end re-read the last pixel, never stored

  // B operand: pixel r, k = 16 s + 8 h .. + 7
  h8 xa[KSTEPS];
  const _Float16* xrow = p.x + mc * K + 8 * h;
#pragma unroll
  for (int s = 0; s < KSTEPS; ++s) xa[s] = ldg16(xrow + 16 * s);

  // A operand rows: MFMA row r <-> channel (of a 64-channel group) perm(r) + 16 j
  const int perm = 32 * ((r >> 2) & 1) + 4 * (r >> 3) + (r & 3);
  const int n_begin = tn * p.nt, n_end = n_begin + p.nt;
  // the shortcut of a group is requested one group AHEAD (two register buffers, the loop handles two groups per
  // trip): its HBM latency is covered by the previous group's weight loads, MFMAs and stores
  h8 ra[4], rb[4];
  if (p.res) {
#pragma unroll
    for (int q = 0; q < 4; ++q) ra[q] = ldg16(p.res + mc * N + n_begin + 32 * h + 8 * q);
  }
  for (int n0 = n_begin; n0 < n_end; n0 += 128) {
    conv1x1_direct_group<KSTEPS>(p, xa, n0, n_end, perm, h, mc, m < p.M, ra, rb);
    if (n0 + 64 < n_end) conv1x1_direct_group<KSTEPS>(p, xa, n0 + 64, n_end, perm, h, mc, m < p.M, rb, ra);
  }
}

extern "C" int odet_conv1x1_f16_direct(const void* x, const void* w, const void* bias, const void* residual, void* y,
                                long long npix, int cin, int cout, int relu, odet_stream_t stream) {
  ODET_REQUIRE(x && w && bias && y, "odet_conv1x1_f16_direct: null pointer");
  ODET_REQUIRE(npix >= 0 && npix < (1ll << 40), "odet_conv1x1_f16_direct: bad pixel count");
  ODET_REQUIRE(cin == 64 || cin == 128 || cin == 256, "odet_conv1x1_f16_direct: input channels must be 64, 128 or 256 (got %d)", cin);
  ODET_REQUIRE(cout > 0 && cout % 64 == 0, "odet_conv1x1_f16_direct: output channels must be a multiple of 64 (got %d)", cout);
  ODET_REQUIRE(((uintptr_t)x | (uintptr_t)w | (uintptr_t)bias | (uintptr_t)residual | (uintptr_t)y) % 16 == 0,
               "odet_conv1x1_f16_direct: pointers must be 16-byte aligned");
  if (npix == 0) return ODET_OK;
  Conv1x1DirectParams p;
  p.x = (const _Float16*)x; p.w = (const _Float16*)w; p.bias = (const _Float16*)bias;
  p.res = (const _Float16*)residual; p.y = (_Float16*)y;
  p.M = npix; p.K = cin; p.N = cout; p.relu = relu ? 1 : 0;
  // channels per wave tile: the whole row up to 256 channels; wider outputs are split so that small feature maps
  // still give the chip enough waves (the waves of a pixel slab sit in one workgroup and share its x lines in L1)
  p.nt = cout <= 256 ? cout : (cout % 256 == 0 ? 256 : 64);
  p.tiles_n = cout / p.nt;
  p.waves = ((npix + 31) / 32) * p.tiles_n;
  const long long blocks = (p.waves + 3) / 4;
  ODET_REQUIRE(blocks < (1ll << 31), "odet_conv1x1_f16_direct: too many workgroups");
  dim3 grid((unsigned)blocks), block(256);
  switch (cin) {
    case 64: hipLaunchKernelGGL(k_conv1x1_f16_direct<4>, grid, block, 0, (hipStream_t)stream, p); break;
    case 128: hipLaunchKernelGGL(k_conv1x1_f16_direct<8>, grid, block, 0, (hipStream_t)stream, p); break;
    default: hipLaunchKernelGGL(k_conv1x1_f16_direct<16>, grid, block, 0, (hipStream_t)stream, p); break;
  }
  ODET_LAUNCH_CHECK();
  return ODET_OK;
}

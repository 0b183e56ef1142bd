// 1x1 stride-1 convolution of the dense path with its whole epilogue, on the matrix cores (SURVEY 8(f) rank 3:
// "Backbone + FPN neck on MFMA ... frozen-BN folded into conv"): the third convolution of every bottleneck block
// (resnet_fpn.py:154-205) and what follows it,
//     y[m, :] = relu( x[m, :] . W^T + bias + shortcut[m, :] )          x [M, K], W [N, K], y / shortcut [M, N]
// in NHWC float16 with float32 accumulation and ONE rounding.  As "library convolution + epilogue pass" the
// [M, N] output makes three more trips through HBM (written by the convolution, read and re-written by the
// epilogue next to the shortcut); these layers are bound by that traffic, not by the contraction (K = 64 .. 256:
// 32 .. 128 FLOP per output byte), so the kernel is built around the streams, not around the MFMA rate:
//
//  * a WORKGROUP (4 waves) owns 128 pixels x NT channels; wave w computes its 32 pixels TRANSPOSED with
//    v_mfma_f32_32x32x16_f16 (D = W_tile . x_tile^T): the MFMA's A operand is 32 rows of W, its B operand 32
//    pixels of x.  A fragment is "8 consecutive k of one row": the x fragments are one 16-byte global load per
//    lane straight into the operand registers and stay there for the whole channel loop (x is read from HBM
//    exactly once); the weights of a 64-channel group ([64, K], <= 32 KB) are staged once per workgroup in LDS
//    (double buffer: the next group travels global -> registers while this one computes, registers -> LDS after
//    it, one barrier per group; rows padded by 16 bytes so that ds_read_b128 of 16 consecutive rows covers all
//    64 banks: SQ_LDS_BANK_CONFLICT = 0).
//  * in the transposed result a lane holds ONE pixel and 16 channels per 32x32 block.  The rows of W are fed in a
//    permuted order (row i of the MFMA = channel 32*((i>>2)&1) + 16*j + 4*(i>>3) + (i&3) of the 64-channel group,
//    j = block 0 / 1), which makes those 16 registers 16 CONSECUTIVE channels and the two blocks of a group 32
//    consecutive channels: the shortcut is read and the output written with 16-byte accesses, 64 contiguous bytes
//    per lane, a full 128-byte line per pixel from the two lane halves.
//  * the shortcut STARTS the accumulators (acc = shortcut, then += W.x) and is requested TWO groups ahead (a
//    wave's 64-channel groups are short next to an HBM round trip); the tile's bias slice sits in LDS; bias +
//    ReLU + one rounding + the store follow the contraction.
//
//  * optionally the PRODUCER's epilogue rides on the operand load (in_bias: x is the preceding 3x3 convolution without
//    bias / ReLU; relu(x + in_bias) is applied to the fragments once per pixel), which removes that layer's pass.
//
// HBM bytes per call: M*K*2 (x) + M*N*2 (shortcut) + M*N*2 (y) -- the algorithmic minimum.  Measured (batch 8 at
// 800x1333; library convolution + epilogue pass beside it): 64 -> 256 at 200x334 146 us = 4.2 TB/s (225 us);
// 128 -> 512 at 100x167 88 us (110 us); 256 -> 1024 at 50x84 53 us (64 us); 256 -> 64 at 200x334 without
// shortcut 78 us (89 us); K = 512 form: 512 -> 2048 over the C4 RoI head's 117 600 pixels 447 us (609 us).  History and the direct-to-register first versions: tools/exp/conv1x1_mfma.hip.
#include <hip/hip_fp16.h>

#include "odet_internal.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

struct Conv1x1Params {
  const _Float16* x; const _Float16* w; const _Float16* bias; const _Float16* res; _Float16* y;
  const _Float16* in_bias;   // nullable: x is a convolution WITHOUT its bias / ReLU; relu(x + in_bias) is applied on load
  long long M;
  int K, N, relu;
  int tiles_n;      // N / nt
  int nt;           // channels per workgroup tile (multiple of 64)
};

__device__ __forceinline__ h8 ldg16(const _Float16* p) { return *reinterpret_cast<const h8*>(p); }

// WORKGROUP tile: 128 pixels (4 waves x 32) x nt channels, walked in groups of 64 channels.  The group's weights
// ([64, K], rows in MFMA order = permuted channels) are staged ONCE per workgroup in LDS (two buffers: the next
// group's rows travel global -> registers while this group computes, registers -> LDS after it, one barrier per
// group); rows are padded by 16 bytes so that the ds_read_b128 of 16 consecutive rows covers all 64 banks.
template <int KSTEPS>
__global__ void __launch_bounds__(256, 2) k_conv1x1_f16(Conv1x1Params p) {
  constexpr int K = 16 * KSTEPS;
  constexpr int LDW = K + 8;                       // LDS row stride in halfs (2K + 16 bytes)
  constexpr int CHUNKS = (64 * K / 8) / 256;       // 16-byte chunks of a weight group per thread: 2, 4, 8
  __shared__ __align__(16) _Float16 wl[2][64 * LDW];
  __shared__ __align__(16) _Float16 bl[256];          // the tile's bias slice (nt <= 256)
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  // XCD-aware order: hardware deals consecutive workgroups round-robin over the 8 XCDs (one L2 each).  The
  // tiles_n channel tiles of a pixel slab read the same x rows, so they are consecutive workgroups of ONE XCD:
  // x comes over the fabric once per slab instead of once per channel tile.
  const long long blk = blockIdx.x;
  const long long q = blk >> 3;
  const long long slab = (blk & 7) + 8 * (q / p.tiles_n);
  const int tn = (int)(q % p.tiles_n);
  if (slab * 128 >= p.M) return;                   // (padding workgroups of the last group of 8 slabs; whole workgroup)
  const int r = lane & 31, h = lane >> 5;
  const int N = p.N;
  const long long m = slab * 128 + wv * 32 + r;
  const long long mc = m < p.M ? m : p.M - 1;      // rows past the end re-read the last pixel, never stored
  const bool store = m < p.M;

  // B operand: pixel r, k = 16 s + 8 h .. + 7 (kept for the whole channel loop: x is read from HBM once)
  h8 xa[KSTEPS];
  const _Float16* xrow = p.x + mc * K + 8 * h;
#pragma unroll
  for (int s = 0; s < KSTEPS; ++s) xa[s] = ldg16(xrow + 16 * s);
  if (p.in_bias) {
    // the producer's epilogue on the way in (the 3x3 convolution before this one ran without bias / ReLU):
    // bias_act's arithmetic -- float32 add, ReLU, one rounding -- on the operand fragments, once per pixel
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
      const h8 bv = ldg16(p.in_bias + 16 * s + 8 * h);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float v = (float)xa[s][e] + (float)bv[e];
        v = (v < 0.0f) ? 0.0f : v;
        xa[s][e] = (_Float16)v;
      }
    }
  }

  const int n_begin = tn * p.nt, n_end = n_begin + p.nt;
  // this thread's chunks of a weight group: LDS row lr (MFMA row lr & 31 of block lr >> 5) <- channel perm
  int goff[CHUNKS], loff[CHUNKS];
#pragma unroll
  for (int i = 0; i < CHUNKS; ++i) {
    const int c = threadIdx.x + 256 * i;
    const int lr = c / (K / 8), cc = c - lr * (K / 8);
    const int ri = lr & 31, j = lr >> 5;
    const int ch = 32 * ((ri >> 2) & 1) + 16 * j + 4 * (ri >> 3) + (ri & 3);
    goff[i] = ch * K + 8 * cc;
    loff[i] = lr * LDW + 8 * cc;
  }
  h8 wreg[CHUNKS];
#pragma unroll
  for (int i = 0; i < CHUNKS; ++i) wreg[i] = ldg16(p.w + (long long)n_begin * K + goff[i]);
  // The shortcut of a group STARTS its accumulators (acc = shortcut, then += W.x) and is requested TWO groups
  // ahead (buffers ra / rb, two groups per loop trip; a buffer is refilled right after its conversion): a wave's
  // groups are short next to an HBM round trip, and with one group of lead every group waited for one.  The
  // tile's bias slice sits in LDS and joins in the epilogue.
  h8 ra[4], rb[4];
  const long long row_off = mc * N + 32 * h;
  if (p.res) {
#pragma unroll
    for (int q = 0; q < 4; ++q) ra[q] = ldg16(p.res + row_off + n_begin + 8 * q);
    if (n_begin + 64 < n_end) {
#pragma unroll
      for (int q = 0; q < 4; ++q) rb[q] = ldg16(p.res + row_off + n_begin + 64 + 8 * q);
    }
  }
  if ((int)threadIdx.x * 8 < p.nt) *reinterpret_cast<h8*>(&bl[threadIdx.x * 8]) = ldg16(p.bias + n_begin + threadIdx.x * 8);
#pragma unroll
  for (int i = 0; i < CHUNKS; ++i) *reinterpret_cast<h8*>(&wl[0][loff[i]]) = wreg[i];
  __syncthreads();

  auto group = [&](const int n0, const int b, h8 (&pre)[4]) {
    const bool more = n0 + 64 < n_end;
    f16v acc0, acc1;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float v = p.res ? (float)pre[q][e] : 0.0f;
        const int idx = (q & 1) * 8 + e;
        if (q < 2) acc0[idx] = v; else acc1[idx] = v;
      }
    }
    if (more) {
#pragma unroll
      for (int i = 0; i < CHUNKS; ++i) wreg[i] = ldg16(p.w + (long long)(n0 + 64) * K + goff[i]);
    }
    if (p.res && n0 + 128 < n_end) {
#pragma unroll
      for (int q = 0; q < 4; ++q) pre[q] = ldg16(p.res + row_off + n0 + 128 + 8 * q);
    }
    const _Float16* l0 = &wl[b][r * LDW + 8 * h];
    const _Float16* l1 = l0 + 32 * LDW;
    // weight fragments two k-steps AHEAD of the MFMAs that use them; the scheduling barriers keep hipcc from
    // sinking the reads back next to their uses (read -> wait -> MFMA per step otherwise)
    h8 f0[3], f1[3];
    f0[0] = *reinterpret_cast<const h8*>(l0); f1[0] = *reinterpret_cast<const h8*>(l1);
    f0[1] = *reinterpret_cast<const h8*>(l0 + 16); f1[1] = *reinterpret_cast<const h8*>(l1 + 16);
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
      if (s + 2 < KSTEPS) {
        f0[(s + 2) % 3] = *reinterpret_cast<const h8*>(l0 + 16 * (s + 2));
        f1[(s + 2) % 3] = *reinterpret_cast<const h8*>(l1 + 16 * (s + 2));
      }
      __builtin_amdgcn_sched_barrier(0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(f0[s % 3], xa[s], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(f1[s % 3], xa[s], acc1, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    // lane (pixel r, half h): channels n0 + 32 h + [0, 32): acc0 -> +0..15, acc1 -> +16..31
    const long long off = mc * N + n0 + 32 * h;
    const _Float16* bp = &bl[n0 - n_begin + 32 * h];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const h8 bv = *reinterpret_cast<const h8*>(bp + 8 * q);
      float fv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int idx = (q & 1) * 8 + e;
        float v = (q < 2 ? acc0[idx] : acc1[idx]) + (float)bv[e];
        if (p.relu) v = (v < 0.0f) ? 0.0f : v;
        fv[e] = v;
      }
      const h8 ov = d_cvt8_f16<h8>(fv);
      if (store) *reinterpret_cast<h8*>(p.y + off + 8 * q) = ov;
    }
    if (more) {
#pragma unroll
      for (int i = 0; i < CHUNKS; ++i) *reinterpret_cast<h8*>(&wl[b ^ 1][loff[i]]) = wreg[i];
    }
    __syncthreads();     // buffer b is free for the group after next; buffer b ^ 1 is complete
  };
  for (int n0 = n_begin; n0 < n_end; n0 += 128) {     // (n_end - n_begin is the same for every wave: uniform trips)
    group(n0, 0, ra);
    if (n0 + 64 < n_end) group(n0 + 64, 1, rb);
  }
}

// K = 512 (a bottleneck's last convolution in conv5: 512 -> 2048): the pixel fragments take 128 registers, so the
// channel loop walks groups of 32 channels (ONE 32x32 MFMA block; a lane owns 16 consecutive channels, the two
// lane halves of a pixel 64 contiguous bytes) and a weight group is [32, 512] = 33 KB per LDS buffer.  Same
// schedule as above: weights double-buffered through LDS, shortcut two groups ahead, bias slice in LDS.
__global__ void __launch_bounds__(256, 2) k_conv1x1_f16_k512(Conv1x1Params p) {
  constexpr int KSTEPS = 32, K = 512, LDW = K + 8;
  constexpr int CHUNKS = (32 * K / 8) / 256;       // 8
  __shared__ __align__(16) _Float16 wl[2][32 * LDW];
  __shared__ __align__(16) _Float16 bl[256];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  // XCD-aware order: hardware deals consecutive workgroups round-robin over the 8 XCDs (one L2 each).  The
  // tiles_n channel tiles of a pixel slab read the same x rows, so they are consecutive workgroups of ONE XCD:
  // x comes over the fabric once per slab instead of once per channel tile.
  const long long blk = blockIdx.x;
  const long long q = blk >> 3;
  const long long slab = (blk & 7) + 8 * (q / p.tiles_n);
  const int tn = (int)(q % p.tiles_n);
  if (slab * 128 >= p.M) return;                   // (padding workgroups of the last group of 8 slabs; whole workgroup)
  const int r = lane & 31, h = lane >> 5;
  const int N = p.N;
  const long long m = slab * 128 + wv * 32 + r;
  const long long mc = m < p.M ? m : p.M - 1;
  const bool store = m < p.M;
  h8 xa[KSTEPS];
  const _Float16* xrow = p.x + mc * K + 8 * h;
#pragma unroll
  for (int s = 0; s < KSTEPS; ++s) xa[s] = ldg16(xrow + 16 * s);
  if (p.in_bias) {
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
      const h8 bv = ldg16(p.in_bias + 16 * s + 8 * h);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float v = (float)xa[s][e] + (float)bv[e];
        v = (v < 0.0f) ? 0.0f : v;
        xa[s][e] = (_Float16)v;
      }
    }
  }
  const int n_begin = tn * p.nt, n_end = n_begin + p.nt;
  // MFMA row i <-> channel 16 * ((i >> 2) & 1) + 4 * (i >> 3) + (i & 3) of the 32-channel group
  int goff[CHUNKS], loff[CHUNKS];
#pragma unroll
  for (int i = 0; i < CHUNKS; ++i) {
    const int c = threadIdx.x + 256 * i;
    const int lr = c / (K / 8), cc = c - lr * (K / 8);
    const int ch = 16 * ((lr >> 2) & 1) + 4 * (lr >> 3) + (lr & 3);
    goff[i] = ch * K + 8 * cc;
    loff[i] = lr * LDW + 8 * cc;
  }
  h8 wreg[CHUNKS];
#pragma unroll
  for (int i = 0; i < CHUNKS; ++i) wreg[i] = ldg16(p.w + (long long)n_begin * K + goff[i]);
  h8 ra[2], rb[2];
  const long long row_off = mc * N + 16 * h;
  if (p.res) {
#pragma unroll
    for (int q = 0; q < 2; ++q) ra[q] = ldg16(p.res + row_off + n_begin + 8 * q);
    if (n_begin + 32 < n_end) {
#pragma unroll
      for (int q = 0; q < 2; ++q) rb[q] = ldg16(p.res + row_off + n_begin + 32 + 8 * q);
    }
  }
  if ((int)threadIdx.x * 8 < p.nt) *reinterpret_cast<h8*>(&bl[threadIdx.x * 8]) = ldg16(p.bias + n_begin + threadIdx.x * 8);
#pragma unroll
  for (int i = 0; i < CHUNKS; ++i) *reinterpret_cast<h8*>(&wl[0][loff[i]]) = wreg[i];
  __syncthreads();

  auto group = [&](const int n0, const int b, h8 (&pre)[2]) {
    const bool more = n0 + 32 < n_end;
    f16v acc;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[8 * q + e] = p.res ? (float)pre[q][e] : 0.0f;
    }
    if (more) {
#pragma unroll
      for (int i = 0; i < CHUNKS; ++i) wreg[i] = ldg16(p.w + (long long)(n0 + 32) * K + goff[i]);
    }
    if (p.res && n0 + 64 < n_end) {
#pragma unroll
      for (int q = 0; q < 2; ++q) pre[q] = ldg16(p.res + row_off + n0 + 64 + 8 * q);
    }
    const _Float16* l0 = &wl[b][r * LDW + 8 * h];
    h8 f0[3];
    f0[0] = *reinterpret_cast<const h8*>(l0);
    f0[1] = *reinterpret_cast<const h8*>(l0 + 16);
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
      if (s + 2 < KSTEPS) f0[(s + 2) % 3] = *reinterpret_cast<const h8*>(l0 + 16 * (s + 2));
      __builtin_amdgcn_sched_barrier(0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(f0[s % 3], xa[s], acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    // lane (pixel r, half h): channels n0 + 16 h + [0, 16)
    const long long off = mc * N + n0 + 16 * h;
    const _Float16* bp = &bl[n0 - n_begin + 16 * h];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const h8 bv = *reinterpret_cast<const h8*>(bp + 8 * q);
      float fv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float v = acc[8 * q + e] + (float)bv[e];
        if (p.relu) v = (v < 0.0f) ? 0.0f : v;
        fv[e] = v;
      }
      const h8 ov = d_cvt8_f16<h8>(fv);
      if (store) *reinterpret_cast<h8*>(p.y + off + 8 * q) = ov;
    }
    if (more) {
#pragma unroll
      for (int i = 0; i < CHUNKS; ++i) *reinterpret_cast<h8*>(&wl[b ^ 1][loff[i]]) = wreg[i];
    }
    __syncthreads();
  };
  for (int n0 = n_begin; n0 < n_end; n0 += 64) {
    group(n0, 0, ra);
    if (n0 + 32 < n_end) group(n0 + 32, 1, rb);
  }
}

extern "C" int odet_conv1x1_f16(const void* x, const void* in_bias, const void* w, const void* bias, const void* residual,
                                void* y, long long npix, int cin, int cout, int relu, odet_stream_t stream) {
  ODET_REQUIRE(x && w && bias && y, "odet_conv1x1_f16: null pointer");
  ODET_REQUIRE(npix >= 0 && npix < (1ll << 40), "odet_conv1x1_f16: bad pixel count");
  ODET_REQUIRE(cin == 64 || cin == 128 || cin == 256 || cin == 512,
               "odet_conv1x1_f16: input channels must be 64, 128, 256 or 512 (got %d)", cin);
  ODET_REQUIRE(cout > 0 && cout % 64 == 0, "odet_conv1x1_f16: output channels must be a multiple of 64 (got %d)", cout);
  ODET_REQUIRE(((uintptr_t)x | (uintptr_t)in_bias | (uintptr_t)w | (uintptr_t)bias | (uintptr_t)residual | (uintptr_t)y) % 16 == 0,
               "odet_conv1x1_f16: pointers must be 16-byte aligned");
  if (npix == 0) return ODET_OK;
  Conv1x1Params p;
  p.x = (const _Float16*)x; p.w = (const _Float16*)w; p.bias = (const _Float16*)bias;
  p.res = (const _Float16*)residual; p.y = (_Float16*)y; p.in_bias = (const _Float16*)in_bias;
  p.M = npix; p.K = cin; p.N = cout; p.relu = relu ? 1 : 0;
  // channels per wave tile: the whole row up to 256 channels; wider outputs are split so that small feature maps
  // still give the chip enough waves (the waves of a pixel slab sit in one workgroup and share its x lines in L1)
  p.nt = cout <= 256 ? cout : (cout % 256 == 0 ? 256 : 64);
  p.tiles_n = cout / p.nt;
  const long long blocks = (((npix + 127) / 128 + 7) / 8) * 8 * p.tiles_n;      // slabs padded to the 8 XCDs
  ODET_REQUIRE(blocks < (1ll << 31), "odet_conv1x1_f16: too many workgroups");
  dim3 grid((unsigned)blocks), block(256);
  switch (cin) {
    case 64: hipLaunchKernelGGL(k_conv1x1_f16<4>, grid, block, 0, (hipStream_t)stream, p); break;
    case 128: hipLaunchKernelGGL(k_conv1x1_f16<8>, grid, block, 0, (hipStream_t)stream, p); break;
    case 512: hipLaunchKernelGGL(k_conv1x1_f16_k512, grid, block, 0, (hipStream_t)stream, p); break;
    default: hipLaunchKernelGGL(k_conv1x1_f16<16>, grid, block, 0, (hipStream_t)stream, p); break;
  }
  ODET_LAUNCH_CHECK();
  return ODET_OK;
}

// Round 6 probe: a 3x3 implicit-GEMM loop with ONE wave per SIMD (4 waves, 256 threads), a 128 x 128 wave tile whose
// accumulators fill the accumulator half of the register file, and the fragment reads of K-step k + 1 issued BETWEEN the
// matrix instructions of K-step k -- the form DESIGN section 8.2 (round 5) names as untried.  float16 NHWC in / out, float32
// accumulation, bias + ReLU; one map, no fused tails: enough to time the loop against k_conv3x3_f16 on the RpnHead's P2 level.
//
//   workgroup tile 256 pixels x 256 channels, K-step = 64 channels of one tap (128-byte rows), two LDS stages of 64 KB
//   wave (wm, wn) of 2 x 2: pixels wm * 128 .., channels wn * 128 ..  -> acc[8][8] tiles of v_mfma_f32_16x16x32_f16 = 256 registers
//   fragments: one K-step = (8 + 8) tiles x 2 K-halves = 32 x ds_read_b128.  The pixel fragments of step k + 1 REPLACE those of
//   step k tile by tile (a tile's 16 MFMAs are issued, then its registers are refilled); the weight fragments are used by every
//   pixel tile of the step, so they are double-buffered (2 x 64 registers).
//   Buffers: at the barrier that ends step k - 1 stage k + 1 has landed and every wave has finished reading stage k (its
//   fragments are in registers): the copies of step k + 2 go into stage k's buffer, the reads of step k take stage k + 1.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef __attribute__((address_space(3))) void* lds_ptr;

struct V2Params {
  const _Float16* x; _Float16* y; const _Float16* w; const _Float16* bias;
  long long M; int H, W, cin, cout, relu, tiles_n; long long slabs;
};

#define V2_TM 256
#define V2_TN 256
#define V2_BK 32                                   // channels per K-step: 64-byte rows
#define V2_NS 4                                    // LDS stages
#define V2_STAGE ((V2_TM + V2_TN) * 64)            // 32 KB

__global__ void __launch_bounds__(256, 1) k_conv3x3_f16_v2(V2Params p) {
  extern __shared__ __align__(16) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wv >> 1, wn = wv & 1;
  const long long blk = blockIdx.x;
  const long long q8 = blk >> 3;
  const long long slab = (blk & 7) + 8 * (q8 / p.tiles_n);
  const int tn = (int)(q8 % p.tiles_n);
  if (slab >= p.slabs) return;
  const int H = p.H, W = p.W, cin = p.cin, cout = p.cout;
  const uint32_t pixB = (uint32_t)cin * 2u, PAD = (uint32_t)(W + 1) * pixB, OOB = 0xFFFFFFF0u;
  const long long M = p.M;
  const rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(p.x)) - PAD, 0,
                                                      (int)((uint32_t)M * pixB + 2u * PAD), 0x00020000);
  const uint32_t wrowB = 9u * pixB;
  const rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.w), 0, (int)((uint32_t)cout * wrowB), 0x00020000);
  // ---- copies: a wave instruction moves 16 rows x 64 B; wave wv takes pieces wv + 4 i (i < 4) of the pixel rows and of the weights.
  // LDS image: 64-byte rows, the 16-byte slot q of rows 8..15 of every 16 stored at q ^ 2 (conflict-free ds_read_b128)
  const int sub = lane >> 2;
  const uint32_t slot = (uint32_t)((lane & 3) ^ ((lane >> 5) << 1)) * 16u;
  uint32_t voffX[4], maskX[4], voffW[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (wv + 4 * i) * 16 + sub;
    const long long m = slab * V2_TM + row;
    uint32_t mk = 0;
    if (m < M) {
      const long long img = m / ((long long)H * W);
      const int rem = (int)(m - img * H * W);
      const int yy = rem / W, xx = rem - yy * W;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int y2 = yy + t / 3 - 1, x2 = xx + t % 3 - 1;
        if (y2 >= 0 && y2 < H && x2 >= 0 && x2 < W) mk |= 1u << t;
      }
    }
    maskX[i] = mk;
    voffX[i] = (uint32_t)m * pixB + slot;
    voffW[i] = (uint32_t)(tn * V2_TN + row) * wrowB + slot;
  }
  const int chunks = cin / V2_BK;
  const int ksteps = 9 * chunks;
  // copy j (0..7) of a wave's 8 per K-step: 0..3 its pixel pieces, 4..7 its weight pieces
  auto issue_piece = [&](int ks, uint32_t stage, int j) {
    if (j < 4) {
      const int tap = ks / chunks, chunk = ks - tap * chunks;
      const uint32_t soX = (uint32_t)((tap / 3) * W + tap % 3) * pixB + (uint32_t)chunk * 64u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr)(lds + stage + (uint32_t)(wv + 4 * j) * 1024u), 16,
                                               (int)(((maskX[j] >> tap) & 1u) ? voffX[j] : OOB), (int)soX, 0, 0);
    } else {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)(lds + stage + (uint32_t)V2_TM * 64u + (uint32_t)(wv + 4 * (j - 4)) * 1024u), 16,
                                               (int)voffW[j - 4], (int)((uint32_t)ks * 64u), 0, 0);
    }
  };
  auto issue = [&](int ks, uint32_t stage) {
#pragma unroll
    for (int j = 0; j < 8; ++j) issue_piece(ks, stage, j);
  };
  // ---- fragments
  const int l15 = lane & 15, lq = lane >> 4;
  const uint32_t fslot = (uint32_t)(lq ^ ((l15 >> 3) << 1)) * 16u;
  const uint32_t xoff = (uint32_t)(wm * 128 + l15) * 64u + fslot;                               // + mt * 1024
  const uint32_t woff = (uint32_t)V2_TM * 64u + (uint32_t)(wn * 128 + l15) * 64u + fslot;     // + t * 1024
  f4 acc[8][8];
#pragma unroll
  for (int mt = 0; mt < 8; ++mt)
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[mt][t] = (f4){0.0f, 0.0f, 0.0f, 0.0f};
  h8 xf[8], wfA[8], wfB[8];
  auto rd = [&](const unsigned char* sb, uint32_t off, int tile) {
    return *reinterpret_cast<const h8*>(sb + off + (uint32_t)tile * 1024u);
  };
  // ---- prologue: steps 0 .. NS - 2 into their buffers (8 copies per wave and step, in order); all fragments of step 0 into registers
#pragma unroll
  for (int j = 0; j < V2_NS - 1; ++j)
    if (j < ksteps) issue(j, (uint32_t)j * (uint32_t)V2_STAGE);
  // (steps 0 and 1 landed; step 2's copies may stay in flight)
  if (ksteps > 2) asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
  for (int t = 0; t < 8; ++t) wfA[t] = rd(lds, woff, t);
#pragma unroll
  for (int mt = 0; mt < 8; ++mt) xf[mt] = rd(lds, xoff, mt);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  // ---- one K-step: the MFMAs of step ks on (wc, xf); behind every pixel tile's 8 MFMAs its registers take the fragment of step
  // ks + 1, and one of the 8 weight fragments of step ks + 1 goes into wn_.  At the start of step ks: stages ks (in registers: its
  // buffer is free), ks + 1 (landed, read during this step), ks + 2 (in flight, awaited at the end of this step); the copies of
  // step ks + 3 go into buffer (ks + 3) % 4 = (ks - 1) % 4, whose fragments were read during step ks - 2.
  auto step = [&](int ks, uint32_t sn, h8 (&wc)[8], h8 (&wn_)[8]) {
    const unsigned char* nb = lds + sn * (uint32_t)V2_STAGE;                    // stage of step ks + 1
#ifdef V2_DIAG_NOREAD
    const bool more = false;
#else
    const bool more = ks + 1 < ksteps;
#endif
#ifdef V2_DIAG_NODMA
    const bool dma = false;
#else
    const bool dma = ks + 3 < ksteps;
#endif
#ifdef V2_BURST
    if (dma) issue(ks + 3, ((sn + 2u) & 3u) * (uint32_t)V2_STAGE);
#endif
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
#ifndef V2_BURST
      // one copy behind every pixel tile's MFMAs: a wave never queues more than one 1-KB piece at the texture path at a time
      // (all 8 at the top of the step stalled the wave's issue -- and with one wave per SIMD the matrix pipe -- behind them)
      if (dma) issue_piece(ks + 3, ((sn + 2u) & 3u) * (uint32_t)V2_STAGE, mt);
#endif
#pragma unroll
      for (int t = 0; t < 8; ++t)      // accumulators pinned to the accumulator file ("a"): the compiler's own MFMA form keeps them in
                                        // VGPRs and shuttles them to AGPRs and back inside the loop once there are 256 of them
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[mt][t]) : "v"(wc[t]), "v"(xf[mt]));
      if (more) {
        xf[mt] = rd(nb, xoff, mt);
        wn_[mt] = rd(nb, woff, mt);
      }
    }
    // stage ks + 2 landed (the copies of ks + 3, issued at the top of this step, may stay in flight)
#if defined(V2_DIAG_NOBAR)
    if (dma) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#elif defined(V2_DIAG_NOSYNC)
#else
    if (dma) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
  };
  int ks = 0;
  uint32_t sn = 1u;                                                              // (ks + 1) % 4
  for (; ks + 1 < ksteps; ks += 2) {
    step(ks, sn, wfA, wfB);
    step(ks + 1, (sn + 1u) & 3u, wfB, wfA);
    sn = (sn + 2u) & 3u;
  }
  if (ks < ksteps) step(ks, sn, wfA, wfB);
  // (an MFMA's result -> any reader but the next MFMA of its chain: wait states the compiler does not know about for an asm MFMA)
#pragma unroll
  for (int mt = 0; mt < 8; ++mt)
    asm volatile("s_nop 15" : "+a"(acc[mt][0]), "+a"(acc[mt][1]), "+a"(acc[mt][2]), "+a"(acc[mt][3]), "+a"(acc[mt][4]), "+a"(acc[mt][5]),
                 "+a"(acc[mt][6]), "+a"(acc[mt][7]));
  // ---- epilogue: lane = pixel l15 of every pixel tile, channels 16 t + 4 lq .. + 3 of the wave's 128
  const int cbase = tn * V2_TN + wn * 128 + lq * 4;
#pragma unroll
  for (int mt = 0; mt < 8; ++mt) {
    const long long m = slab * V2_TM + wm * 128 + mt * 16 + l15;
    if (m < M) {
      _Float16* dst = p.y + m * cout + cbase;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        h4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float v = acc[mt][t][j] + (p.bias ? (float)p.bias[cbase + 16 * t + j] : 0.0f);
          if (p.relu) v = v < 0.0f ? 0.0f : v;
          o[j] = (_Float16)v;
        }
        *reinterpret_cast<h4*>(dst + 16 * t) = o;
      }
    }
  }
}

extern "C" int v2_conv3x3_f16(const void* x, const void* w, const void* bias, void* y, int batch, int H, int W, int cin, int cout,
                              int relu, void* stream) {
  if (cin % 32 || cout % 256) return -1;
  static bool once = false;
  if (!once) {
    if (hipFuncSetAttribute((const void*)k_conv3x3_f16_v2, hipFuncAttributeMaxDynamicSharedMemorySize, V2_NS * V2_STAGE) != hipSuccess) return -2;
    once = true;
  }
  V2Params p;
  p.x = (const _Float16*)x; p.y = (_Float16*)y; p.w = (const _Float16*)w; p.bias = (const _Float16*)bias;
  p.M = (long long)batch * H * W; p.H = H; p.W = W; p.cin = cin; p.cout = cout; p.relu = relu;
  p.tiles_n = cout / V2_TN;
  p.slabs = (p.M + V2_TM - 1) / V2_TM;
  if ((unsigned long long)p.M * cin * 2ull + 2ull * (W + 1) * cin * 2ull >= 0xFFFFFFF0ull) return -3;
  const long long blocks = (p.slabs + 7) / 8 * 8 * p.tiles_n;
  hipLaunchKernelGGL(k_conv3x3_f16_v2, dim3((unsigned)blocks), dim3(256), V2_NS * V2_STAGE, (hipStream_t)stream, p);
  return hipGetLastError() == hipSuccess ? 0 : -4;
}

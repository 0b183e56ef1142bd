// The float32 forms of the implicit-GEMM convolution kernel (conv3x3.hip) -- the detectors' PARITY mode computes in the
// reference's precision, so its dense layers run on exact-float32 matrix instructions (v_mfma_f32_16x16x4_f32: every
// product and every sum rounded to float32 once, a chain of fmaf; 1/16 of the float16 rate = the float32 vector rate):
//
//   k_conv3x3_f32<MT, WN>    3x3 stride-1 'same' convolutions (resnet_fpn.py:154-205 the bottlenecks' middle convolution,
//                            :399-407 the neck's smoothing convolutions, base_fpn_model.py:401-417 the RpnHead) + bias + ReLU
//   k_pointwise_f32<MT, WN>  1x1 convolutions (stride 1 or 2) and dense layers: the bottlenecks' first / last / shortcut
//                            convolutions (last + shortcut of a stage's first block as ONE contraction along K), the neck's
//                            P5 and lateral convolutions (top-down merge in the epilogue, resnet_fpn.py:385-398), the RoI
//                            head's Dense layers (resnet_fpn.py:292-336), the stem's 7x7 / 2 convolution on its patch matrix
//                            (k_stem_patches_f32) -- shortcut / merge / ReLU in the epilogue
//
// Same tiling, LDS-DMA staging (128-byte rows, 16-byte slots XOR-swizzled by the row) and transposed-tile layout as the
// float16 kernel with 4-byte elements: a K-step is (one tap x) 32 input channels; a lane reads the two 16-byte slots q and
// 4 + q of its row (q = lane >> 4) and uses float j of those eight in MFMA step j -- both operands order K the same way, so
// every k of the 32 is visited exactly once.  The loop is bound by the matrix pipe (no pipelining finesse needed): 135
// TFLOP/s on the RpnHead's P2 level = the float32 matrix rate at the clock the chip holds.  Workgroup tile: 256 / 128 / 64
// channels x 128..256 pixels, picked per launch; two LDS stages of the tile's own size.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <mutex>

#include "conv_f32_common.h"

typedef __amdgpu_buffer_rsrc_t f32_rsrc_t;
typedef __attribute__((address_space(3))) void* f32_lds_ptr;

#define F32_BK 32                 // input channels per K-step (128 bytes per row)
#define F32_LDS_MAX (160 * 1024)

template <int MT, int WN, int TAPS>
__device__ __forceinline__ void conv_tile_f32(const ConvF32Params& p) {
  constexpr int WM = 8 / WN;
  constexpr int TM = WM * 16 * MT;
  constexpr int TN = 64 * WN;
  constexpr int XP = (TM / 8 + 7) / 8;
  constexpr uint32_t STAGE = (uint32_t)(TM + TN) * 128u;
  constexpr uint32_t WBASE = (uint32_t)TM * 128u;
  // small tiles run a THREE-stage ring (two K-steps of copies in flight): a layer with few pixels leaves one or two
  // workgroups per CU, and with two stages their matrix pipe waits for every copy (conv5's 3x3 at batch 4: 305 -> ~190 us)
  constexpr int NSTAGE = (TM % 64 == 0 && 3 * (TM + TN) * 128 <= 160 * 1024) ? 3 : 2;
  constexpr int LOADS = TM / 64 + WN;                    // copies per wave and K-step (uniform when TM % 64 == 0)
  extern __shared__ __align__(16) unsigned char lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wv / WN, wn = wv % WN;
  const long long blk = blockIdx.x;
  const long long q8 = blk >> 3;
  const long long slab = (blk & 7) + 8 * (q8 / p.tiles_n);
  const int tn = (int)(q8 % p.tiles_n);
  if (slab >= p.tile_start[p.num_levels]) return;
  int lv = 0;
#pragma unroll
  for (int l = 1; l < ODET_MAX_LEVELS; ++l)
    if (l < p.num_levels && slab >= p.tile_start[l]) lv = l;
  const long long tile_m = slab - p.tile_start[lv];
  const int H = p.H[lv], W = p.W[lv], cin = p.cin, cout = p.cout;
  const uint32_t pixB = (uint32_t)cin * 4u;
  const uint32_t PAD = TAPS == 9 ? (uint32_t)(W + 1) * pixB : 0u;
  const uint32_t OOB = 0xFFFFFFF0u;
  const long long M = p.M[lv];
  const long long Min = TAPS == 9 ? M : p.Min;
  const f32_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(p.x[lv])) - PAD, 0, (int)((uint32_t)Min * pixB + 2u * PAD), 0x00020000);
  const bool dual = TAPS == 1 && p.x2 != nullptr;
  const uint32_t pixB2 = dual ? (uint32_t)p.cin2 * 4u : 0u;
  const f32_rsrc_t rx2 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(dual ? p.x2 : p.x[lv])), 0, (int)((uint32_t)(dual ? p.Min2 : 0) * pixB2), 0x00020000);
  const uint32_t wrowB = (uint32_t)TAPS * pixB + pixB2;
  const f32_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, (int)((uint32_t)cout * wrowB), 0x00020000);
  const int sub = lane >> 3;
  const uint32_t slot = (uint32_t)((lane & 7) ^ sub) * 16u;
  uint32_t voffA[XP], voffA2[TAPS == 1 ? XP : 1], voffW[WN], maskA[XP];
#pragma unroll
  for (int i = 0; i < XP; ++i) {
    const int row = (wv + 8 * i) * 8 + sub;
    const long long m = tile_m * TM + row;
    uint32_t mk = 0;
    if constexpr (TAPS == 9) {
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
      voffA[i] = (uint32_t)m * pixB + slot;
    } else {
      long long src = m;
      if (p.stride != 1 && m < M) {
        const long long opx = (long long)p.Ho * p.Wo;
        const long long img = m / opx;
        const int rem = (int)(m - img * opx);
        const int yo = rem / p.Wo, xo = rem - yo * p.Wo;
        src = (img * H + (long long)yo * p.stride) * W + (long long)xo * p.stride;
      }
      mk = m < M ? 1u : 0u;
      voffA[i] = (uint32_t)(dual ? m : src) * pixB + slot;
      voffA2[i] = (uint32_t)src * pixB2 + slot;
    }
    maskA[i] = mk;
  }
#pragma unroll
  for (int i = 0; i < WN; ++i) {
    const int row = (wv * WN + i) * 8 + sub;
    // LDS row rho of the W tile <- channel rho (identity): an MFMA tile t of a wave's 64-channel group then leaves lane
    // (pixel, q) with channels 16 t + 4 q .. + 3, i.e. the four lanes of a pixel hold 64 CONTIGUOUS bytes of the output row
    // per tile -- one store instruction writes whole 64-byte segments (the float16 kernel's permuted order, 16 consecutive
    // channels per lane, would scatter float32 results in 16-byte pieces 64 bytes apart)
    const int ch = tn * TN + row;
    voffW[i] = (uint32_t)ch * wrowB + slot;
  }
  const int chunks = cin / F32_BK;
  const int ksteps = TAPS * chunks + (dual ? p.cin2 / F32_BK : 0);
  const int k1steps = dual ? p.k1steps : ksteps;
  auto issue = [&](int ks, uint32_t stage) {
    int tap = 0;
    uint32_t soA;
    if constexpr (TAPS == 9) {
      tap = ks / chunks;
      const int chunk = ks - tap * chunks;
      soA = (uint32_t)((tap / 3) * W + tap % 3) * pixB + (uint32_t)chunk * 128u;
    } else {
      soA = (uint32_t)ks * 128u;
    }
    if (TAPS == 1 && ks >= k1steps) {
#pragma unroll
      for (int i = 0; i < XP; ++i) {
        if ((wv + 8 * i) * 8 < TM) {
          const uint32_t va = (maskA[i] & 1u) ? voffA2[i] : OOB;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rx2, (f32_lds_ptr)(lds + stage + (uint32_t)(wv + 8 * i) * 1024u), 16, (int)va,
                                                   (int)((uint32_t)(ks - k1steps) * 128u), 0, 0);
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < XP; ++i) {
        if ((wv + 8 * i) * 8 < TM) {
          const uint32_t va = ((maskA[i] >> tap) & 1u) ? voffA[i] : OOB;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (f32_lds_ptr)(lds + stage + (uint32_t)(wv + 8 * i) * 1024u), 16, (int)va,
                                                   (int)soA, 0, 0);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < WN; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (f32_lds_ptr)(lds + stage + WBASE + (uint32_t)(wv * WN + i) * 1024u), 16,
                                               (int)voffW[i], (int)((uint32_t)ks * 128u), 0, 0);
  };
  const int l15 = lane & 15, lq = lane >> 4;
  const uint32_t fslot = (uint32_t)(lq ^ (lane & 7)) * 16u;                         // slot q; slot 4 + q = ^ 64
  const uint32_t xoff = (uint32_t)(wm * 16 * MT + l15) * 128u + fslot;
  const uint32_t woff = WBASE + (uint32_t)(wn * 64 + l15) * 128u + fslot;
  c3f4 acc[MT][4];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[mt][t] = (c3f4){0.0f, 0.0f, 0.0f, 0.0f};
  auto compute = [&](const unsigned char* sb) {
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
      const uint32_t kx = kh ? 64u : 0u;
      c3f4 wf[4], xf[MT];
#pragma unroll
      for (int t = 0; t < 4; ++t) wf[t] = *reinterpret_cast<const c3f4*>(sb + ((woff + (uint32_t)t * 2048u) ^ kx));
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) xf[mt] = *reinterpret_cast<const c3f4*>(sb + ((xoff + (uint32_t)mt * 2048u) ^ kx));
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int t = 0; t < 4; ++t)
            acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[t][j], xf[mt][j], acc[mt][t], 0, 0, 0);
    }
  };
  if constexpr (NSTAGE == 3) {
    issue(0, 0u);
    if (ksteps > 1) issue(1, STAGE);
    uint32_t cur = 0u;                                     // stage of K-step ks; ks + 2 goes where ks - 1 was read
    for (int ks = 0; ks < ksteps; ++ks) {
      if (ks + 1 < ksteps) {
        // all but my LOADS newest copies (= K-step ks + 1) have landed
        if constexpr (LOADS == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else if constexpr (LOADS == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if constexpr (LOADS == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else if constexpr (LOADS == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __syncthreads();
      const uint32_t prev = cur == 0u ? 2u * STAGE : cur - STAGE;
      if (ks + 2 < ksteps) issue(ks + 2, prev);
      compute(lds + cur);
      cur = cur == 2u * STAGE ? 0u : cur + STAGE;
    }
  } else {
    issue(0, 0u);
    for (int ks = 0; ks < ksteps; ++ks) {
      const uint32_t stage = (uint32_t)(ks & 1) * STAGE;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (ks + 1 < ksteps) issue(ks + 1, STAGE - stage);
      compute(lds + stage);
    }
  }
  conv_f32_epilogue<MT, TAPS>(p, acc, tile_m, TM, TN, wm, wn, tn, l15, lq, lv, M, cout);
}

template <int MT, int WN>
__global__ void __launch_bounds__(512) k_conv3x3_f32(ConvF32Params p) {
  conv_tile_f32<MT, WN, 9>(p);
}

template <int MT, int WN>
__global__ void __launch_bounds__(512) k_pointwise_f32(ConvF32Params p) {
  conv_tile_f32<MT, WN, 1>(p);
}

// ---- host side --------------------------------------------------------------------------------------------------------------
template <typename F>
static void f32_for_each_kernel(F f) {
  f((const void*)k_conv3x3_f32<4, 4>); f((const void*)k_conv3x3_f32<5, 4>); f((const void*)k_conv3x3_f32<6, 4>);
  f((const void*)k_conv3x3_f32<7, 4>); f((const void*)k_conv3x3_f32<8, 4>); f((const void*)k_conv3x3_f32<2, 2>);
  f((const void*)k_conv3x3_f32<3, 2>); f((const void*)k_conv3x3_f32<4, 2>); f((const void*)k_conv3x3_f32<1, 1>);
  f((const void*)k_conv3x3_f32<2, 1>);
  f((const void*)k_pointwise_f32<4, 4>); f((const void*)k_pointwise_f32<5, 4>); f((const void*)k_pointwise_f32<6, 4>);
  f((const void*)k_pointwise_f32<7, 4>); f((const void*)k_pointwise_f32<8, 4>); f((const void*)k_pointwise_f32<2, 2>);
  f((const void*)k_pointwise_f32<3, 2>); f((const void*)k_pointwise_f32<4, 2>); f((const void*)k_pointwise_f32<1, 1>);
  f((const void*)k_pointwise_f32<2, 1>);
}

static hipError_t f32_prepare_kernels() {
  static OdetPerDeviceOnce once;
  return once.run([] {
    hipError_t rc = hipSuccess;
    f32_for_each_kernel([&rc](const void* k) {
      const hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, F32_LDS_MAX);
      if (e != hipSuccess) rc = e;
    });
    return rc;
  });
}

static unsigned f32_lds_bytes(int tm, int tn) {          // (the kernel's NSTAGE rule)
  const int stages = (tm % 64 == 0 && 3 * (tm + tn) * 128 <= 160 * 1024) ? 3 : 2;
  return (unsigned)(stages * (tm + tn) * 128);
}

// tile (channels 64 * wn, pixels (8 / wn) * 16 * mt) with the least rounds x time per K-step: the float32 matrix pipe needs
// TM * TN / 2 cycles per K-step of 32 (16 x the float16 kernel's), staging 2 * (TM + TN): always matrix-bound, so only the
// fill of the rounds of 256 workgroups (two per CU for tiles of <= 80 KB) and the per-workgroup overhead matter
static void f32_pick_tile(const long long* M, int num_levels, int cout, int* wn_out, int* mt_out) {
  int wn_best = 0, mt_best = 0;
  double best = 1e300;
  for (int wn = 4; wn >= 1; wn >>= 1) {
    if (cout % (64 * wn)) continue;
    const int wm = 8 / wn, tn = 64 * wn, tiles_n = cout / tn;
    for (int mt = 16 / wm; mt >= 8 / wm; --mt) {
      const int tm = wm * 16 * mt;
      long long slabs = 0;
      for (int l = 0; l < num_levels; ++l) slabs += (M[l] + tm - 1) / tm;
      const long long blocks = (slabs + 7) / 8 * 8 * tiles_n;
      const int lds = (int)f32_lds_bytes(tm, tn);
      const int occ = std::max(1, std::min(3, (160 * 1024) / lds));
      // matrix cycles per K-step of the tile, + what a workgroup loses per K-step to its barrier and copy latency when
      // nothing else runs on the CU (measured, tools/exp/f32_layers.py: the 128 x 64 tile wins or ties on every layer)
      const double per = (double)tm * tn / 2.0 + 1500.0 / occ;
      const double cost = (double)((blocks + 256 * occ - 1) / (256 * occ)) * per * occ;
      if (cost < best * 0.98) { best = cost; wn_best = wn; mt_best = mt; }
    }
  }
  *wn_out = wn_best; *mt_out = mt_best;
}

template <bool PW>
static void f32_launch_tile(int wn, int mt, dim3 grid, unsigned lds_bytes, hipStream_t st, const ConvF32Params& p) {
#define F32_L(MT_, WN_)                                                                                     \
  do {                                                                                                      \
    if (PW) hipLaunchKernelGGL((k_pointwise_f32<MT_, WN_>), grid, dim3(512), lds_bytes, st, p);             \
    else hipLaunchKernelGGL((k_conv3x3_f32<MT_, WN_>), grid, dim3(512), lds_bytes, st, p);                  \
  } while (0)
  switch (wn * 16 + mt) {
    case 4 * 16 + 4: F32_L(4, 4); break;
    case 4 * 16 + 5: F32_L(5, 4); break;
    case 4 * 16 + 6: F32_L(6, 4); break;
    case 4 * 16 + 7: F32_L(7, 4); break;
    case 4 * 16 + 8: F32_L(8, 4); break;
    case 2 * 16 + 2: F32_L(2, 2); break;
    case 2 * 16 + 3: F32_L(3, 2); break;
    case 2 * 16 + 4: F32_L(4, 2); break;
    case 1 * 16 + 1: F32_L(1, 1); break;
    default: F32_L(2, 1); break;
  }
#undef F32_L
}

static void f32_defaults(ConvF32Params* p) {
  p->stride = 1; p->Ho = p->Wo = 0; p->Min = 0; p->res = nullptr; p->top = nullptr; p->th = p->tw = 0; p->tys = p->txs = 0.0f;
  p->x2 = nullptr; p->cin2 = 0; p->k1steps = 0; p->Min2 = 0;
  p->ksplit = 0; p->part = nullptr; p->ticket = nullptr; p->acc_scale = 1.0f; p->status = nullptr;
}

static int conv3x3_f32_launch(const odet_conv_level_t* levels, int num_levels, const void* w, const void* bias, int batch,
                              int cin, int cout, int relu, hipStream_t st) {
  ODET_REQUIRE(levels && w, "odet_conv3x3_f32: null pointer");
  ODET_REQUIRE(num_levels >= 1 && num_levels <= ODET_MAX_LEVELS, "odet_conv3x3_f32: num_levels %d out of range", num_levels);
  ODET_REQUIRE(batch > 0, "odet_conv3x3_f32: bad batch");
  ODET_REQUIRE(cin > 0 && cin % F32_BK == 0, "odet_conv3x3_f32: cin %d must be a multiple of %d", cin, F32_BK);
  ODET_REQUIRE(cout > 0 && cout % 64 == 0, "odet_conv3x3_f32: cout %d must be a multiple of 64", cout);
  ODET_REQUIRE((unsigned long long)cout * 9ull * cin * 4ull < 0x7FFFFFFFull, "odet_conv3x3_f32: weights too large");
  ODET_HIP(f32_prepare_kernels());
  ConvF32Params p;
  f32_defaults(&p);
  for (int l = 0; l < ODET_MAX_LEVELS; ++l) {
    const odet_conv_level_t& L = levels[l < num_levels ? l : 0];
    ODET_REQUIRE(L.x && L.y && L.H > 0 && L.W > 0, "odet_conv3x3_f32: bad level %d", l);
    const long long M = (long long)batch * L.H * L.W;
    ODET_REQUIRE((unsigned long long)M * cin * 4ull + 2ull * (L.W + 1) * cin * 4ull < 0xFFFFFFF0ull,
                 "odet_conv3x3_f32: level %d input larger than 4 GiB", l);
    p.x[l] = (const float*)L.x; p.y[l] = (float*)L.y; p.M[l] = M; p.H[l] = L.H; p.W[l] = L.W;
  }
  int wn, mt;
  f32_pick_tile(p.M, num_levels, cout, &wn, &mt);
  const int TMsel = (8 / wn) * 16 * mt;
  long long total = 0;
  for (int l = 0; l < ODET_MAX_LEVELS; ++l) {
    p.tile_start[l] = total;
    if (l < num_levels) total += (p.M[l] + TMsel - 1) / TMsel;
  }
  for (int l = num_levels; l <= ODET_MAX_LEVELS; ++l) p.tile_start[l] = total;
  p.w = (const float*)w; p.bias = (const float*)bias;
  p.num_levels = num_levels; p.cin = cin; p.cout = cout; p.relu = relu ? 1 : 0;
  p.tiles_n = cout / (64 * wn);
  const long long blocks = (total + 7) / 8 * 8 * p.tiles_n;
  ODET_REQUIRE(blocks < (1ll << 31), "odet_conv3x3_f32: too many workgroups");
  f32_launch_tile<false>(wn, mt, dim3((unsigned)blocks), f32_lds_bytes(TMsel, 64 * wn), st, p);
  ODET_LAUNCH_CHECK();
  return ODET_OK;
}

extern "C" int odet_conv3x3_f32(const void* x, const void* w, const void* bias, void* y, int batch, int H, int W, int cin,
                                int cout, int relu, odet_stream_t stream) {
  ODET_REQUIRE(x && y, "odet_conv3x3_f32: null pointer");
  const odet_conv_level_t one{x, y, H, W};
  return conv3x3_f32_launch(&one, 1, w, bias, batch, cin, cout, relu, (hipStream_t)stream);
}

extern "C" int odet_conv3x3_f32_levels(const odet_conv_level_t* levels, int num_levels, const void* w, const void* bias,
                                       int batch, int cin, int cout, int relu, odet_stream_t stream) {
  return conv3x3_f32_launch(levels, num_levels, w, bias, batch, cin, cout, relu, (hipStream_t)stream);
}

struct PwF32Epilogue { const void* res; const void* top; int th, tw; const void* x2; int cin2; };

static int pointwise_f32_launch(const char* who, const void* x, const void* w, const void* bias, void* y, int batch, int H,
                                int W, int stride, int cin, int cout, int relu, const PwF32Epilogue& epi, hipStream_t st) {
  ODET_REQUIRE(x && w && y, "%s: null pointer", who);
  ODET_REQUIRE(batch > 0 && H > 0 && W > 0 && (stride == 1 || stride == 2), "%s: bad shape", who);
  ODET_REQUIRE(cin % F32_BK == 0 && cin + (epi.x2 ? epi.cin2 : 0) >= 2 * F32_BK,
               "%s: cin %d must be a multiple of %d, at least %d along K", who, cin, F32_BK, 2 * F32_BK);
  ODET_REQUIRE(cout > 0 && cout % 64 == 0, "%s: cout %d must be a multiple of 64", who, cout);
  ODET_REQUIRE(((uintptr_t)x | (uintptr_t)w | (uintptr_t)y | (uintptr_t)bias | (uintptr_t)epi.res | (uintptr_t)epi.top |
                (uintptr_t)epi.x2) % 16 == 0, "%s: pointers must be 16-byte aligned", who);
  ODET_REQUIRE(!(epi.res && epi.top), "%s: shortcut and top-down merge exclude each other", who);
  ODET_REQUIRE(!epi.top || (stride == 1 && epi.th > 0 && epi.tw > 0 && !relu), "%s: bad merge arguments", who);
  ODET_HIP(f32_prepare_kernels());
  const int Ho = (H + stride - 1) / stride, Wo = (W + stride - 1) / stride;
  const long long M = (long long)batch * Ho * Wo;
  const long long Min = epi.x2 ? M : (long long)batch * H * W;
  ODET_REQUIRE((unsigned long long)Min * cin * 4ull < 0xFFFFFFF0ull, "%s: input larger than 4 GiB", who);
  ODET_REQUIRE(!epi.x2 || (epi.cin2 > 0 && epi.cin2 % F32_BK == 0 &&
                           (unsigned long long)batch * H * W * epi.cin2 * 4ull < 0xFFFFFFF0ull), "%s: bad second source", who);
  ODET_REQUIRE((unsigned long long)cout * (cin + (epi.x2 ? epi.cin2 : 0)) * 4ull < 0x7FFFFFFFull, "%s: weights too large", who);
  ConvF32Params p;
  f32_defaults(&p);
  for (int l = 0; l < ODET_MAX_LEVELS; ++l) {
    p.x[l] = (const float*)x; p.y[l] = (float*)y; p.M[l] = M; p.H[l] = H; p.W[l] = W;
  }
  p.res = (const float*)epi.res;
  p.top = (const float*)epi.top; p.th = epi.th; p.tw = epi.tw;
  p.tys = epi.top ? (float)epi.th / (float)Ho : 0.0f;
  p.txs = epi.top ? (float)epi.tw / (float)Wo : 0.0f;
  p.stride = stride; p.Ho = Ho; p.Wo = Wo; p.Min = Min;
  p.x2 = (const float*)epi.x2; p.cin2 = epi.x2 ? epi.cin2 : 0; p.k1steps = cin / F32_BK; p.Min2 = (long long)batch * H * W;
  int wn, mt;
  f32_pick_tile(&M, 1, cout, &wn, &mt);
  const int TMsel = (8 / wn) * 16 * mt;
  p.tiles_n = cout / (64 * wn);
  const long long total = (M + TMsel - 1) / TMsel;
  p.tile_start[0] = 0;
  for (int l = 1; l <= ODET_MAX_LEVELS; ++l) p.tile_start[l] = total;
  p.w = (const float*)w; p.bias = (const float*)bias;
  p.num_levels = 1; p.cin = cin; p.cout = cout; p.relu = relu ? 1 : 0;
  const long long blocks = (total + 7) / 8 * 8 * p.tiles_n;
  ODET_REQUIRE(blocks < (1ll << 31), "%s: too many workgroups", who);
  f32_launch_tile<true>(wn, mt, dim3((unsigned)blocks), f32_lds_bytes(TMsel, 64 * wn), st, p);
  ODET_LAUNCH_CHECK();
  return ODET_OK;
}

extern "C" int odet_pointwise_f32(const void* x, const void* w, const void* bias, const void* residual, void* y, int batch,
                                  int H, int W, int stride, int cin, int cout, int relu, odet_stream_t stream) {
  const PwF32Epilogue e{residual, nullptr, 0, 0, nullptr, 0};
  return pointwise_f32_launch("odet_pointwise_f32", x, w, bias, y, batch, H, W, stride, cin, cout, relu, e, (hipStream_t)stream);
}

extern "C" int odet_lateral_merge_f32(const void* x, const void* w, const void* bias, const void* top, int th, int tw, void* y,
                                      int batch, int H, int W, int cin, int cout, odet_stream_t stream) {
  ODET_REQUIRE(top, "odet_lateral_merge_f32: null pointer");
  const PwF32Epilogue e{nullptr, top, th, tw, nullptr, 0};
  return pointwise_f32_launch("odet_lateral_merge_f32", x, w, bias, y, batch, H, W, 1, cin, cout, 0, e, (hipStream_t)stream);
}

extern "C" int odet_pointwise_dual_f32(const void* x1, int cin1, const void* x2, int cin2, int H2, int W2, int stride2,
                                       const void* w, const void* bias, void* y, int batch, int cout, int relu,
                                       odet_stream_t stream) {
  ODET_REQUIRE(x2, "odet_pointwise_dual_f32: null pointer");
  const PwF32Epilogue e{nullptr, nullptr, 0, 0, x2, cin2};
  return pointwise_f32_launch("odet_pointwise_dual_f32", x1, w, bias, y, batch, H2, W2, stride2, cin1, cout, relu, e,
                              (hipStream_t)stream);
}

// ---- the stem's patch matrix (float32 mode): conv1_pad + Conv2D(64, 7x7, strides 2, 'valid') (resnet_fpn.py:262-289) as a
// GEMM needs its operand rows materialised once -- row m = (image, yo, xo) holds the 7 x 7 x 3 window at (2 yo - 3, 2 xo - 3),
// zero outside the image, in (dy, dx, channel) order (= the keras kernel's HWIO order), zero-padded from 147 to 160 floats.
__global__ void __launch_bounds__(256) k_stem_patches_f32(const float* __restrict__ img, float* __restrict__ out, int B, int H,
                                                          int W, int Ho, int Wo) {
  const long long total = (long long)B * Ho * Wo * 40;          // 40 float4 per row
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int q = (int)(i % 40);
    long long m = i / 40;
    const int xo = (int)(m % Wo);
    m /= Wo;
    const int yo = (int)(m % Ho);
    const int b = (int)(m / Ho);
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k = q * 4 + e;
      float val = 0.0f;
      if (k < 147) {
        const int dy = k / 21, r = k - dy * 21, dx = r / 3, c = r - dx * 3;
        const int yy = 2 * yo - 3 + dy, xx = 2 * xo - 3 + dx;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) val = img[(((long long)b * H + yy) * W + xx) * 3 + c];
      }
      v[e] = val;
    }
    reinterpret_cast<float4*>(out)[i] = make_float4(v[0], v[1], v[2], v[3]);
  }
}

extern "C" int odet_stem_patches_f32(const float* images, float* patches, int batch, int H, int W, odet_stream_t stream) {
  ODET_REQUIRE(images && patches && batch > 0 && H > 0 && W > 0, "odet_stem_patches_f32: bad arguments");
  const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
  const long long total = (long long)batch * Ho * Wo * 40;
  const int grid = (int)std::min<long long>((total + 255) / 256, 256 * 64);
  hipLaunchKernelGGL(k_stem_patches_f32, dim3(grid), dim3(256), 0, (hipStream_t)stream, images, patches, batch, H, W, Ho, Wo);
  ODET_LAUNCH_CHECK();
  return ODET_OK;
}

// ---- VGG16's first convolution in the float32 mode (vgg16_faster_rcnn.py:260-342: Conv2D(64, 3x3, 'same') on the 3-channel
// image) the same way: row m = (image, y, x) holds the 3 x 3 x 3 window at (y - 1, x - 1), zero outside the image, in
// (dy, dx, channel) order, zero-padded from 27 to 64 floats (two K-steps of the pointwise GEMM).
__global__ void __launch_bounds__(256) k_rgb_patches3x3_f32(const float* __restrict__ img, float* __restrict__ out, int B, int H,
                                                            int W) {
  const long long total = (long long)B * H * W * 16;            // 16 float4 per row
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int q = (int)(i & 15);
    long long m = i >> 4;
    const int x = (int)(m % W);
    m /= W;
    const int y = (int)(m % H);
    const int b = (int)(m / H);
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k = q * 4 + e;
      float val = 0.0f;
      if (k < 27) {
        const int dy = k / 9, r = k - dy * 9, dx = r / 3, c = r - dx * 3;
        const int yy = y - 1 + dy, xx = x - 1 + dx;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) val = img[(((long long)b * H + yy) * W + xx) * 3 + c];
      }
      v[e] = val;
    }
    reinterpret_cast<float4*>(out)[i] = make_float4(v[0], v[1], v[2], v[3]);
  }
}

extern "C" int odet_rgb_patches3x3_f32(const float* images, float* patches, int batch, int H, int W, odet_stream_t stream) {
  ODET_REQUIRE(images && patches && batch > 0 && H > 0 && W > 0, "odet_rgb_patches3x3_f32: bad arguments");
  const long long total = (long long)batch * H * W * 16;
  const int grid = (int)std::min<long long>((total + 255) / 256, 256 * 64);
  hipLaunchKernelGGL(k_rgb_patches3x3_f32, dim3(grid), dim3(256), 0, (hipStream_t)stream, images, patches, batch, H, W);
  ODET_LAUNCH_CHECK();
  return ODET_OK;
}

// 3x3 stride-1 'same' convolution as an implicit GEMM on the matrix cores -- the RPN head's convolution
// (model/fpn/base_fpn_model.py:401-417: Conv2D(512, 3x3, padding='same') on every pyramid level,
// 106 GMAC = 31 % of the FLOPs of a ResNet-101-FPN pass), NHWC float16 in / out, float32 accumulation:
//
//     y[p, n] = sum_{tap = (dy, dx), c}  x[p + (dy-1, dx-1), c] . w[n, tap, c]          (zero outside the map)
//
// i.e. a GEMM  Y[M, N] = A[M, K] . W[N, K]^T  with M = B*H*W pixels, N = cout, K = 9 * cin, whose A operand is never
// materialised: row p of the K-step (tap, 64-channel chunk) is the 128-byte line of pixel p + tap offset.
//
//  * WORKGROUP = 8 waves, tile 256 pixels x 256 channels, K-step 64 (one tap, 64 input channels); wave (wm, wn) of
//    2 x 4 owns 128 pixels x 64 channels = 8 x 4 tiles of v_mfma_f32_16x16x32_f16 (128 accumulator registers).
//  * Both operands travel global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds: no registers, no ds_write), whole
//    128-byte lines: a wave instruction moves 8 rows x 128 B.  The LDS image is row-major [256 rows][128 B] with the
//    16-byte slot XOR-swizzled by (row & 7), applied on the SOURCE address (the DMA writes lane-linear) and on the
//    ds_read_b128 address: conflict-free fragment reads.  Zero padding costs nothing: a lane whose tap falls outside
//    the map gets an out-of-range buffer offset, and an out-of-range buffer load returns 0.
//  * Two LDS stages (2 x 64 KB): the DMA of K-step k+1 is in flight while step k computes; one barrier per K-step.
//  * TRANSPOSED tiles (D = W_tile . X_tile^T: the MFMA's A operand is 16 rows of W, its B operand 16 pixels): a lane
//    then holds ONE pixel and 4 channels per tile.  The rows of W are laid out in LDS in a permuted order (LDS row
//    16 t + r of a wave's 64-channel group = channel 16 (r >> 2) + 4 t + (r & 3)) which makes the 16 registers of a
//    pixel 16 CONSECUTIVE channels (the fused forms, whose second GEMM consumes them in K order), or -- plain and
//    pointwise forms -- in the order that gives a lane channels 8 q .. 8 q + 7 of both 32-channel halves, so that each of
//    its two 16-byte stores joins the other three lanes of the pixel in 64 contiguous bytes (round 3: 16-byte pieces
//    32 bytes apart cost the output-heavy 1x1 layers 10-20 %).
//  * XCD-aware order: the channel tiles of a pixel slab are consecutive workgroups of ONE XCD (its L2 serves their
//    common input lines).
//
// Optional epilogue: + bias, ReLU (float32, one rounding).  The RPN head runs it WITHOUT: its tail kernel
// (rpn_tail.hip) applies bias + ReLU to its operand fragments on load.
#include <hip/hip_fp16.h>

#include <cstdio>
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <mutex>

#include "odet_internal.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t c3_rsrc_t;
typedef __attribute__((address_space(3))) void* c3_lds_ptr;

#define C3_TM 256                 // pixels per workgroup tile
#define C3_TN 256                 // channels per workgroup tile
#define C3_EARLY_WAVES 4          // waves below this issue their copies right after the barrier, the others half a step later
#define C3_BK 64                  // input channels per K-step (128 bytes per row)
#define C3_STAGE_BYTES ((C3_TM + C3_TN) * C3_BK * 2)      // 64 KB
#define C3_LDS_BYTES (2 * C3_STAGE_BYTES)

struct Conv3x3Params {           // up to ODET_MAX_LEVELS maps (the pyramid levels the RpnHead is shared by) in ONE launch
  const _Float16* x[ODET_MAX_LEVELS]; _Float16* y[ODET_MAX_LEVELS];
  const _Float16* w; const _Float16* bias;
  long long M[ODET_MAX_LEVELS];   // batch * H * W of a level
  int H[ODET_MAX_LEVELS], W[ODET_MAX_LEVELS];
  long long tile_start[ODET_MAX_LEVELS + 1];   // first pixel slab of a level; [num_levels] = total
  int num_levels, cin, cout, relu;
  int tiles_n;                    // cout / 256
  // fused RpnHead tail (k_conv3x3_f16<.., true>): the 1x1 score / delta convolutions on relu(conv + bias), written
  // straight into the concatenated float32 arrays -- the 512-channel activation never goes to memory
  const _Float16* tail_w;         // [6A][cout]: 2A score rows, then 4A delta rows
  const _Float16* tail_b;         // [6A]
  float* scores; float* deltas;   // [batch][N][2] / [batch][N][4]
  // fused bottleneck tail (k_conv3x3_f16<.., false, true>): y3 = relu(relu(conv + bias) . w3^T + b3 + res), cout == 256
  const _Float16* w3; const _Float16* b3; const _Float16* res; _Float16* y3;
  int n3, relu3;
  long long s_stride, d_stride;   // values per image
  long long px[ODET_MAX_LEVELS];  // H * W of a level
  long long aoff[ODET_MAX_LEVELS];   // first anchor of a level inside an image
  int A;
  // pointwise form (k_pointwise_f16: TAPS == 1, one map): a 1x1 convolution / dense layer, optionally strided --
  // output row m = (image, yo, xo) of a Ho x Wo map reads input pixel (yo * stride, xo * stride) of the H x W map
  int stride, Ho, Wo;
  long long Min;                  // input rows (batch * H * W)
  // its epilogues: + shortcut `res` [M][cout] (shared with the fused tail's member), or the FPN top-down merge
  // (resnet_fpn.py:385-398): out = 0.5 * resize_bilinear(top) + 0.5 * (conv + bias), top [batch][th][tw][cout]
  const _Float16* top; int th, tw; float tys, txs;
  // or float32 results (the network's last layer: class logits / box regressions leave in float32): y32 [M][cout], bias32
  float* y32; const float* bias32;
  // or TWO sources along K (a stage's first bottleneck, resnet_fpn.py:154-205: last 1x1 convolution + convolutional
  // shortcut + Add + ReLU as ONE contraction over [y2 | x(::stride)] with the weights concatenated): K-steps below k1steps
  // read x (rows m, cin channels), the others x2 (rows of the strided H x W map, cin2 channels)
  const _Float16* x2; int cin2, k1steps; long long Min2;
  // plain 3x3 form with the 2x2 / 2 'same' max-pooling of VGG16's stages in its epilogue (vgg16_faster_rcnn.py:260-342): the
  // launch's pixel index runs over (image, row PAIR, column, row in pair) of the map rounded up to even sizes, so that four
  // consecutive indices are one pooling window (the four lanes of a DPP quad) and index >> 2 is the pooled pixel; M counts
  // those indices, Min the real pixels; y is the pooled map [batch][ceil(H/2)][ceil(W/2)][cout]
  int pool;
};

// WN = waves along the channels (4: 256-channel tile, the form described above; 2 / 1: 128 / 64-channel tiles for the
// layers with fewer output channels, the other 8 / WN waves along the pixels); MT = 16-pixel tiles per wave: the
// workgroup tile is TM = (8 / WN) * 16 * MT pixels x TN = 64 * WN channels (WN = 4, MT = 8: 256 x 256; smaller MT for
// launches whose slabs would fill a fraction of a round of the 256 CUs -- chosen on the host, conv3x3_launch).
//
// NW = waves of the workgroup (8; 4 for the SMALL-M tiles: 64 pixels x 64 / 128 channels), NS = LDS stages.  NS == 2 is the
// half-step-pipelined loop described above, tuned for launches whose tiles keep the matrix pipe busy.  NS > 2 is the RING
// form for launches with few pixels (batch 1 .. 4 on the 50 x 84 and 25 x 42 maps: the BASELINE configs' own batch size):
// there a launch is a few hundred small workgroups, each K-step's matrix work (64 x 64 x 64: ~130 cycles per SIMD) is far
// shorter than the round trip of its copies, and with two stages every step would wait for one.  The ring keeps NS - 1
// K-steps of copies in flight (counted vmcnt waits: the LDS-DMA pieces of a wave land in issue order), one barrier per
// step: the copy of step ks + NS - 1 goes where step ks - 1 was read.
template <int N>
__device__ __forceinline__ void c3_wait_vmcnt_barrier() {
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}

template <int MT, int WN, bool TAIL, bool BLK, int TAPS, int NW = 8, int NS = 2>
__device__ __forceinline__ void conv_tile_f16(const Conv3x3Params& p) {
  static_assert(TAPS == 9 || (TAPS == 1 && !TAIL && !BLK), "taps");
  static_assert((NW == 8 || NW == 4) && NW % WN == 0 && (8 * WN) % NW == 0, "waves");
  static_assert(NS >= 2 && (NS == 2 || (!TAIL && !BLK)), "stages");
  constexpr bool LIN = !TAIL && !BLK;                    // the order of a lane's output channels, see voffW below
  constexpr int WM = NW / WN;                            // waves along the pixels
  constexpr int TM = WM * 16 * MT;                       // pixels of the workgroup tile (<= 256)
  constexpr int TN = 64 * WN;                            // channels of the workgroup tile
  constexpr int XP = (TM / 8 + NW - 1) / NW;             // pixel pieces (8 rows x 128 B) per wave: TM / 8 over the waves
  constexpr int WPW = 8 * WN / NW;                       // weight pieces per wave: TN / 8 over the waves
  static_assert(TM <= C3_TM && TM % 8 == 0, "tile");
  // a stage = the pixel rows, then the weight rows, 128 bytes each: (TM + TN) * 128 bytes (64 KB for the 256 x 256 tile;
  // small tiles leave room for a second workgroup on the CU -- the launch asks for 2 stages of its own tile)
  constexpr uint32_t STAGE = (uint32_t)(TM + TN) * 128u;
  constexpr uint32_t WBASE = (uint32_t)TM * 128u;
  extern __shared__ __align__(16) unsigned char lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wv / WN, wn = wv % WN;
  // workgroup -> (pixel slab, channel tile): the channel tiles of a slab on one XCD
  const long long blk = blockIdx.x;
  const long long q8 = blk >> 3;
  // (fused RpnHead: one workgroup per slab, its channel tiles one after the other -- see the TAIL section)
  const int tiles_grid = TAIL ? 1 : p.tiles_n;
  const long long slab = (blk & 7) + 8 * (q8 / tiles_grid);
  int tn = (int)(q8 % tiles_grid);
  if (slab >= p.tile_start[p.num_levels]) return;        // (padding workgroups of the last group of 8 slabs)
  int lv = 0;                                            // the level this slab belongs to (big levels first: the small
#pragma unroll                                           //  ones fill the tail of the launch)
  for (int l = 1; l < ODET_MAX_LEVELS; ++l)
    if (l < p.num_levels && slab >= p.tile_start[l]) lv = l;
  const long long tile_m = slab - p.tile_start[lv];
  const int H = p.H[lv], W = p.W[lv], cin = p.cin, cout = p.cout;
  const uint32_t pixB = (uint32_t)cin * 2u;              // bytes per pixel
  const uint32_t PAD = TAPS == 9 ? (uint32_t)(W + 1) * pixB : 0u;   // the descriptor starts one row + one pixel before x
  const uint32_t OOB = 0xFFFFFFF0u;
  const long long M = p.M[lv];
  const long long Min = (TAPS == 9 && !p.pool) ? M : p.Min;   // input rows (a strided pointwise layer reads more than it writes)
  const c3_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(p.x[lv])) - PAD, 0, (int)((uint32_t)Min * pixB + 2u * PAD), 0x00020000);
  const bool dual = TAPS == 1 && p.x2 != nullptr;
  const uint32_t pixB2 = dual ? (uint32_t)p.cin2 * 2u : 0u;
  const c3_rsrc_t rx2 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(dual ? p.x2 : p.x[lv])), 0, (int)((uint32_t)(dual ? p.Min2 : 0) * pixB2), 0x00020000);
  const uint32_t wrowB = (uint32_t)TAPS * pixB + pixB2;  // bytes per weight row [tap][cin] (+ [cin2] of the second source)
  const c3_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.w), 0, (int)((uint32_t)cout * wrowB),
                                                         0x00020000);
  // ---- what this thread copies per K-step: its pieces of A (8 pixels x 128 B each; piece wv + 8 i) and 4 of W
  const int sub = lane >> 3;                             // row of the 8-row piece
  const uint32_t slot = (uint32_t)((lane & 7) ^ sub) * 16u;   // logical 16-byte slot this lane fetches (XOR swizzle)
  uint32_t voffA[XP], voffW[WPW];
  uint32_t voffA2[TAPS == 1 ? XP : 1];                   // the second source's rows (pointwise form with two sources)
  uint32_t maskA[XP];                                    // bit tap: the tap of this lane's pixel is inside the map
#pragma unroll
  for (int i = 0; i < XP; ++i) {
    const int row = (wv + NW * i) * 8 + sub;             // 0..TM-1 (pieces beyond the tile are never issued)
    const long long m = tile_m * TM + row;
    uint32_t mk = 0;
    if constexpr (TAPS == 9) {
      long long src = m;
      if (m < M) {
        long long img;
        int yy, xx;
        bool inside = true;
        if (p.pool) {                                      // (image, row pair, column, row in pair) of the even-rounded map
          const int Wp = (W + 1) & ~1, Hp = (H + 1) & ~1;
          const long long Sp = (long long)Hp * Wp;
          img = m / Sp;
          const int rem = (int)(m - img * Sp);
          const int pair = rem / (2 * Wp), r2 = rem - pair * 2 * Wp;
          xx = r2 >> 1; yy = 2 * pair + (r2 & 1);
          inside = yy < H && xx < W;
          src = inside ? (img * H + yy) * W + xx : 0;
        } else {
          img = m / ((long long)H * W);
          const int rem = (int)(m - img * H * W);
          yy = rem / W; xx = rem - yy * W;
        }
        if (inside) {
#pragma unroll
          for (int t = 0; t < 9; ++t) {
            const int y2 = yy + t / 3 - 1, x2 = xx + t % 3 - 1;
            if (y2 >= 0 && y2 < H && x2 >= 0 && x2 < W) mk |= 1u << t;
          }
        }
      }
      voffA[i] = (uint32_t)src * pixB + slot;            // (+ the tap / chunk offset as soffset; PAD is in the base)
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
  for (int tn_pass = 0; tn_pass < (TAIL ? p.tiles_n : 1); ++tn_pass) {
  if (TAIL) tn = tn_pass;
#pragma unroll
  for (int i = 0; i < WPW; ++i) {
    const int row = (wv * WPW + i) * 8 + sub;            // 0..TN-1
    // LDS row rho of the W tile <- channel.  The fused forms (their second GEMM takes a lane's 16 results as K-ordered
    // operands): rows 16 t + r of a 64-channel group hold channel 16 (r >> 2) + 4 t + (r & 3) -- 16 consecutive channels
    // per lane.  The plain and pointwise forms (LIN): channel 32 (t >> 1) + 8 (r >> 2) + 4 (t & 1) + (r & 3) -- a lane
    // holds channels 8 q .. 8 q + 7 of each 32-channel half, so the four lanes of a pixel store 64 CONTIGUOUS bytes per
    // instruction instead of four 16-byte pieces 32 bytes apart
    const int g = row >> 6, rr = row & 63, t = rr >> 4, r = rr & 15;
    const int ch = tn * TN + g * 64 + (LIN ? 32 * (t >> 1) + 8 * (r >> 2) + 4 * (t & 1) + (r & 3) : 16 * (r >> 2) + 4 * t + (r & 3));
    voffW[i] = (uint32_t)ch * wrowB + slot;
  }
  const int chunks = cin / C3_BK;
  const int ksteps = TAPS * chunks + (dual ? p.cin2 / C3_BK : 0);
  const int k1steps = dual ? p.k1steps : ksteps;
  struct IssueAt { int tap; uint32_t soA, soW, stage; };
  auto issue_at = [&](int ks, uint32_t stage) {
    IssueAt a;
    if constexpr (TAPS == 9) {
      a.tap = ks / chunks;
      const int chunk = ks - a.tap * chunks;
      a.soA = (uint32_t)((a.tap / 3) * W + a.tap % 3) * pixB + (uint32_t)chunk * 128u;
    } else {
      a.tap = 0;
      a.soA = (uint32_t)ks * 128u;
    }
    a.soW = (uint32_t)ks * 128u;
    a.stage = stage;
    return a;
  };
  auto issue = [&](int ks, uint32_t stage) {              // 1 KB pieces: this wave's share of the pixel rows + of the weights
    const IssueAt a = issue_at(ks, stage);
    // (ring form: the steps past the end are issued all the same, out of range -- no traffic, zeros into a stage nobody
    // reads any more -- so that every step leaves the same number of copies in flight for the counted waits)
    const bool live = NS == 2 || ks < ksteps;
    if (TAPS == 1 && ks >= k1steps) {                      // (wave-uniform) the second source's K-steps
#pragma unroll
      for (int i = 0; i < XP; ++i) {
        if ((TM / 8) % NW == 0 || (wv + NW * i) * 8 < TM) {
          const uint32_t va = ((maskA[i] & 1u) && live) ? voffA2[i] : OOB;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rx2, (c3_lds_ptr)(lds + a.stage + (uint32_t)(wv + NW * i) * 1024u), 16, (int)va,
                                                   (int)((uint32_t)(ks - k1steps) * 128u), 0, 0);
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < XP; ++i) {
        if ((TM / 8) % NW == 0 || (wv + NW * i) * 8 < TM) {   // (TM / 8 not a multiple of NW: the last piece exists for the first waves only)
          const uint32_t va = (((maskA[i] >> a.tap) & 1u) && live) ? voffA[i] : OOB;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (c3_lds_ptr)(lds + a.stage + (uint32_t)(wv + NW * i) * 1024u), 16, (int)va,
                                                   (int)a.soA, 0, 0);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < WPW; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (c3_lds_ptr)(lds + a.stage + WBASE + (uint32_t)(wv * WPW + i) * 1024u), 16,
                                               (int)(live ? voffW[i] : OOB), (int)a.soW, 0, 0);
  };
  // ---- fragment addresses (bytes inside a stage)
  const int l15 = lane & 15, lq = lane >> 4;
  const uint32_t fslot = (uint32_t)(lq ^ (lane & 7)) * 16u;                         // K half 0; half 1 = ^ 64
  const uint32_t xoff = (uint32_t)(wm * 16 * MT + l15) * 128u + fslot;              // + mt * 2048
  const uint32_t woff = WBASE + (uint32_t)(wn * 64 + l15) * 128u + fslot;    // + t * 2048
  f4 acc[MT][4];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[mt][t] = (f4){0.0f, 0.0f, 0.0f, 0.0f};

  // K loop, software-pipelined by half a K-step: the fragments of (ks, half 1) are read and the MFMAs of (ks, half 0)
  // issued BEFORE the barrier that publishes stage ks + 1, so the matrix pipe has half a step of work queued while the
  // waves meet, the next DMA is issued and the first fragments of step ks + 1 come out of LDS -- the burst of LDS reads
  // right after a barrier no longer leaves the MFMAs waiting.
  auto read_frags = [&](const unsigned char* sb, uint32_t kx, h8 (&wf)[4], h8 (&xf)[MT]) {
#pragma unroll
    for (int t = 0; t < 4; ++t) wf[t] = *reinterpret_cast<const h8*>(sb + ((woff + (uint32_t)t * 2048u) ^ kx));
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) xf[mt] = *reinterpret_cast<const h8*>(sb + ((xoff + (uint32_t)mt * 2048u) ^ kx));
  };
  auto mfmas = [&](const h8 (&wf)[4], const h8 (&xf)[MT]) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[t], xf[mt], acc[mt][t], 0, 0, 0);
  };
  const int c0 = tn * TN + wn * 64 + lq * (LIN ? 8 : 16);
  constexpr int HOFF = LIN ? 32 : 8;                     // channel offset of a lane's second group of 8 results
  // the plain epilogue's bias slice, requested before the K loop (two 16-byte loads; a short launch cannot afford to wait for
  // them after it)
  h8 bh[2];
#pragma unroll
  for (int e = 0; e < 8; ++e) bh[0][e] = bh[1][e] = (_Float16)0.0f;
  if (LIN && p.bias) {
    bh[0] = *reinterpret_cast<const h8*>(p.bias + c0);
    bh[1] = *reinterpret_cast<const h8*>(p.bias + c0 + HOFF);
  }
  h8 wf0[4], xf0[MT], wf1[4], xf1[MT];
  if constexpr (NS > 2) {
    // ---- ring form (small-M launches): NS - 1 K-steps of copies in flight, one barrier per step
    static_assert((TM / 8) % NW == 0, "ring form: every wave issues the same number of copies per step");
    constexpr int PW = XP + WPW;                         // copies per wave and K-step
    constexpr int INFLIGHT = (NS - 2) * PW;              // the copies of the NS - 2 steps younger than the one about to be read
    static_assert(INFLIGHT <= 63, "vmcnt");
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) issue(s, (uint32_t)s * STAGE);
    uint32_t cur = 0u, fill = (uint32_t)(NS - 1) * STAGE;
    for (int ks = 0; ks < ksteps; ++ks) {
      // my copies of step ks have landed; after the barrier everybody's, and everybody has read step ks - 1 (its fragments
      // went into MFMAs issued before this point), whose stage is refilled now.  A bare s_barrier: __syncthreads() would
      // make the compiler drain ALL copies in flight (vmcnt(0)) -- the ring's whole point is that they stay in flight
      c3_wait_vmcnt_barrier<INFLIGHT>();
      issue(ks + NS - 1, fill);
      read_frags(lds + cur, 0u, wf0, xf0);
      read_frags(lds + cur, 64u, wf1, xf1);
      mfmas(wf0, xf0);
      mfmas(wf1, xf1);
      fill = cur;
      cur = cur + STAGE == (uint32_t)NS * STAGE ? 0u : cur + STAGE;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // (the out-of-range copies of the last steps)
  } else {
  issue(0, 0u);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  issue(1, STAGE);                              // (ksteps >= 2: nine taps, or cin >= 128 in the pointwise form)
  read_frags(lds, 0u, wf0, xf0);
  // steady state: no conditionals around MFMAs or the barrier (hipcc would wait for the NEW fragment reads before the
  // queued MFMAs of the old ones -- the bubble the half-step pipelining is there to remove); the last two K-steps are
  // peeled.  STAGGER: the two waves of a SIMD (w and w + 4) run the same program between the same barriers, so they
  // would reach their copy-issue phase (8 LDS-DMA pieces, ~100 cycles of issue each, no MFMA of that wave meanwhile)
  // together and leave the matrix pipe idle; waves 0-3 issue right after the barrier, waves 4-7 after the half step's
  // MFMAs (wave-uniform branches around the copies only): -4.3 % on the grouped RPN-head launch.  Measured and
  // rejected placements: tools/exp/conv3x3_issue_placement.patch (all late -2 %, pieces spread between the MFMAs +8 %,
  // early after the fragment reads or in mid-half +5 %).
  int ks = 0;
  const bool early = wv < C3_EARLY_WAVES;
  for (; ks + 2 < ksteps; ++ks) {
    const uint32_t cur = (uint32_t)(ks & 1) * STAGE, nxt = STAGE - cur;
    read_frags(lds + cur, 64u, wf1, xf1);
    mfmas(wf0, xf0);
    // my copies of step ks + 1 have landed and my reads of stage ks are done; after the barrier everybody's
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (early) issue(ks + 2, cur);
    read_frags(lds + nxt, 0u, wf0, xf0);
    mfmas(wf1, xf1);
    if (!early) issue(ks + 2, cur);
  }
  {   // step ksteps - 2: nothing left to issue
    const uint32_t cur = (uint32_t)(ks & 1) * STAGE, nxt = STAGE - cur;
    read_frags(lds + cur, 64u, wf1, xf1);
    mfmas(wf0, xf0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    read_frags(lds + nxt, 0u, wf0, xf0);
    mfmas(wf1, xf1);
    // step ksteps - 1
    read_frags(lds + nxt, 64u, wf1, xf1);
    mfmas(wf0, xf0);
    mfmas(wf1, xf1);
  }
  }
  if constexpr (TAIL) {
    // ---- fused RpnHead tail (base_fpn_model.py:401-434): t = relu(conv + b1) rounded to float16 once, then the two 1x1
    // convolutions as ONE more contraction on the matrix cores, out[o][pixel] = sum_ch W2[o][ch] . t[pixel][ch]: the MFMA's
    // A operand = 16 rows of W2 (o = 2A score rows, then 4A delta rows, zero rows up to 32), its B operand = the 16 pixels
    // of a tile; a lane's 16 channels are two K groups of 8 (the K order inside an MFMA is free as long as both operands
    // agree).  A wave has 64 of the tile's 256 channels: the four waves along the channels add up through LDS (the
    // stages are free after the K loop); the workgroup walks the channel tiles of its slab (two for 512 channels) one
    // after the other and keeps the sums in registers (round 4; before, every channel tile had its own workgroup and
    // left partial sums in a workspace for a second launch -- 2 GB and 225 us at 30 images).
    static_assert(WN == 4, "the fused tail is for 256-channel tiles");
    // (the lane's and the wave's coordinates again, opaque: or everything below that does not depend on the channel tile --
    // LDS addresses, the pixel's image / offset division, the output pointers -- is hoisted out of the channel-tile loop
    // and has to live through the K loop, which spills)
    int lane_o = lane, wv_o = wv;
    asm volatile("" : "+v"(lane_o), "+s"(wv_o));
    const int lane = lane_o, l15 = lane_o & 15, lq = lane_o >> 4, wm = wv_o / WN, wn = wv_o % WN;
    const int rows = 6 * p.A;
    h8 w2[2][2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const int R = rt * 16 + l15;
        h8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (_Float16)0.0f;
        if (R < rows) v = *reinterpret_cast<const h8*>(p.tail_w + (long long)R * cout + c0 + hh * 8);
        w2[rt][hh] = v;
      }
    float b1[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) b1[e] = (float)p.bias[c0 + e];
    __syncthreads();                                     // every wave has read its last fragments: the stages are free
    f4* red = reinterpret_cast<f4*>(lds);                // [wm][mt][rt][wn][lane]
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      h8 t0, t1;
      float tf[16];
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float v = acc[mt][t][j] + b1[t * 4 + j];
          tf[t * 4 + j] = v < 0.0f ? 0.0f : v;
        }
      t0 = d_cvt8_f16<h8>(tf); t1 = d_cvt8_f16<h8>(tf + 8);
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        f4 o = (f4){0.0f, 0.0f, 0.0f, 0.0f};
        o = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2[rt][0], t0, o, 0, 0, 0);
        o = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2[rt][1], t1, o, 0, 0, 0);
        red[(((wm * MT + mt) * 2 + rt) * 4 + wn) * 64 + lane] = o;
      }
    }
    __syncthreads();
    // wave (wm, wn) adds up the pixel tiles mt = wn, wn + 4 of its half (lane = rows 4 lq .. 4 lq + 3 of a row tile,
    // pixel l15) over the four waves along the channels, and over the channel tiles of the slab (this loop): the sums so
    // far wait in the 128 bytes per pixel of LDS behind the stages (a lane reads back what it wrote itself; registers
    // would have to survive the K loop, which has none to spare).  The last channel tile adds the bias and writes row R of
    // pixel m to scores (R < 2A) / deltas (R - 2A) of the level's slice of the concatenated arrays
    // (base_fpn_model.py:188-200, 427-432).
    f4* keep = reinterpret_cast<f4*>(lds + 2u * STAGE);  // [wm][mt][rt][lane]
    const bool first = tn_pass == 0, last = tn_pass + 1 == p.tiles_n;
    const int nS = 2 * p.A;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const int mt = wn + 4 * s2;
      if (mt < MT) {
        const long long m = tile_m * TM + wm * 16 * MT + mt * 16 + l15;
        const long long b = m / p.px[lv], pi = m - b * p.px[lv];
        float* so = p.scores + b * p.s_stride + (p.aoff[lv] + pi * p.A) * 2;
        float* dO = p.deltas + b * p.d_stride + (p.aoff[lv] + pi * p.A) * 4;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
          const f4* src = red + (((wm * MT + mt) * 2 + rt) * 4) * 64 + lane;
          f4* kp = keep + ((wm * MT + mt) * 2 + rt) * 64 + lane;
          const f4 a0 = src[0], a1 = src[64], a2 = src[128], a3 = src[192];
          f4 v;
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = ((a0[j] + a1[j]) + a2[j]) + a3[j];
          if (!first) v += *kp;
          if (!last) {
            *kp = v;
          } else if (m < M) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int R = rt * 16 + 4 * lq + j;
              if (R < rows) {
                const float o = v[j] + (float)p.tail_b[R];
                if (R < nS) so[R] = o; else dO[R - nS] = o;
              }
            }
          }
        }
      }
    }
    if (last) return;
    __syncthreads();                                     // (the next channel tile's copies overwrite `red`)
    continue;
  }
  if constexpr (BLK) {
    // ---- fused bottleneck tail (resnet_fpn.py:154-205: conv 3x3 -> BN -> ReLU -> conv 1x1 -> BN -> Add -> ReLU with the
    // frozen batch norms folded): this workgroup holds ALL CMID = 64 * WN channels of the 3x3 convolution for its pixels
    // (256: ResNet's conv4, 128: conv3, 64: conv2), so the block's last convolution can run right here.  t = relu(acc + bias)
    // goes to LDS once, rounded to float16 ([TM pixels][2 CMID bytes], 16-byte slots XOR-swizzled by the pixel row:
    // conflict-free as an MFMA operand), then every wave takes 64-channel groups of the n3 output channels for ALL pixels of
    // the tile: its 4 x (CMID / 32) weight fragments straight from global memory into registers (rows in the permuted
    // order that leaves a pixel's four lanes with 64 contiguous bytes per store), per pixel tile CMID / 32 fragment reads + 4 CMID / 32 MFMAs,
    // + bias + shortcut, ReLU, one rounding, two 16-byte stores per lane.  The activation of the 3x3 convolution never goes
    // to memory.
    constexpr int CMID = 64 * WN;                        // channels of the 3x3 convolution = K of the 1x1 convolution
    constexpr int KS3 = CMID / 32;                       // its K-steps
    constexpr uint32_t TROW = (uint32_t)CMID * 2u;       // bytes of a pixel's row of t
    constexpr uint32_t SMASK = (uint32_t)(CMID / 8 - 1) < 15u ? (uint32_t)(CMID / 8 - 1) : 15u;   // slots swizzled among
    float b2v[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) b2v[e] = (float)p.bias[c0 + e];
    __syncthreads();                                     // every wave has read its last fragments: the stages are free
    auto taddr = [&](int prow, int slot) -> uint32_t {
      return (uint32_t)prow * TROW + (uint32_t)((((uint32_t)slot & SMASK) ^ ((uint32_t)prow & SMASK)) | ((uint32_t)slot & ~SMASK)) * 16u;
    };
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      h8 t0, t1;
      float tf[16];
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float v = acc[mt][t][j] + b2v[t * 4 + j];
          tf[t * 4 + j] = v < 0.0f ? 0.0f : v;
        }
      t0 = d_cvt8_f16<h8>(tf); t1 = d_cvt8_f16<h8>(tf + 8);
      const int prow = wm * 16 * MT + mt * 16 + l15;
      const int slot0 = wn * 8 + lq * 2;
      *reinterpret_cast<h8*>(lds + taddr(prow, slot0)) = t0;
      *reinterpret_cast<h8*>(lds + taddr(prow, slot0 + 1)) = t1;
    }
    __syncthreads();
    const int n3 = p.n3;
    for (int g = wv; g * 64 < n3; g += NW) {
      h8 a[4][KS3];
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
        const int ch = g * 64 + 32 * (tt >> 1) + 8 * (l15 >> 2) + 4 * (tt & 1) + (l15 & 3);
        const _Float16* wr = p.w3 + (long long)ch * CMID + lq * 8;
#pragma unroll
        for (int ks = 0; ks < KS3; ++ks) a[tt][ks] = *reinterpret_cast<const h8*>(wr + ks * 32);
      }
      const int cg = g * 64 + lq * 8;                    // this lane's output channels: cg .. cg + 7 and cg + 32 .. cg + 39
      float b3v[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) b3v[e] = (float)p.b3[cg + (e >> 3) * 32 + (e & 7)];
      // The weight fragments must have ARRIVED before the pixel loop: the compiler puts the wait for a load at its first
      // use, which is inside the loop -- and a `s_waitcnt vmcnt(0)` there also waits, on every step, for the previous
      // step's output stores (the counter is in order on gfx950).  These empty statements use the registers here.
#pragma unroll
      for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int ks = 0; ks < KS3; ++ks) asm volatile("" :: "v"(a[tt][ks]));
      // The shortcut rows of a later step are requested BEFORE this step's stores for the same reason (waiting for them
      // then leaves the younger stores in flight).  Loads and stores go through buffer descriptors WITHOUT branches: a row
      // past the end (or no shortcut at all: an empty descriptor) reads 0 and its store is dropped by the range check --
      // behind a branch the compiler can no longer count the operations in flight and falls back to vmcnt(0).
      constexpr int NPT = TM / 16;
      static_assert(NPT % 2 == 0, "pixel tiles");
      typedef unsigned int u4 __attribute__((ext_vector_type(4)));
      const uint32_t rowB = (uint32_t)n3 * 2u;
      const c3_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.res ? p.res : p.y3), 0,
                                                               p.res ? (int)((uint32_t)M * rowB) : 0, 0x00020000);
      const c3_rsrc_t ry3 = __builtin_amdgcn_make_buffer_rsrc(p.y3, 0, (int)((uint32_t)M * rowB), 0x00020000);
      const uint32_t off0 = (uint32_t)(tile_m * TM + l15) * rowB + (uint32_t)cg * 2u;   // pixel tile 0 (M * rowB < 2^32: host)
      auto fetch = [&](int pt, u4 (&d)[2]) {
        const long long m = tile_m * TM + pt * 16 + l15;
        const uint32_t off = (m < M && pt < NPT) ? off0 + (uint32_t)pt * 16u * rowB : OOB;   // (pt >= NPT: the last steps' "next")
        d[0] = __builtin_amdgcn_raw_buffer_load_b128(rres, (int)off, 0, 0);
        d[1] = __builtin_amdgcn_raw_buffer_load_b128(rres, (int)off, 64, 0);
      };
      // TWO steps of lead (a step is ~0.5 us for the two waves of a SIMD, a loaded HBM round trip 1-2 us): four register
      // buffers in rotation, the loop unrolled by four so that the rotation is a renaming.  The dropped stores (range check)
      // give the loop's entry the same sequence of operations in flight as its back edge, so the compiler's count at the
      // head of the loop is the steady-state one instead of vmcnt(0).
      u4 r0[2], r1[2], r2[2], r3[2];
      auto drop2 = [&]() {
        __builtin_amdgcn_raw_buffer_store_b128((u4){0u, 0u, 0u, 0u}, ry3, (int)OOB, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128((u4){0u, 0u, 0u, 0u}, ry3, (int)OOB, 64, 0);
      };
      fetch(0, r0); drop2();
      fetch(1, r1); drop2();
      auto step = [&](int pt, u4 (&cur)[2], u4 (&nxt)[2]) {
        fetch(pt + 2, nxt);
        __builtin_amdgcn_sched_barrier(0);
        const int prow = pt * 16 + l15;
        const long long m = tile_m * TM + prow;
        h8 bf[KS3];
#pragma unroll
        for (int ks = 0; ks < KS3; ++ks) bf[ks] = *reinterpret_cast<const h8*>(lds + taddr(prow, 4 * ks + lq));
        f4 o[4];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) o[tt] = (f4){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int ks = 0; ks < KS3; ++ks)                 // four independent accumulation chains, interleaved
#pragma unroll
          for (int tt = 0; tt < 4; ++tt) o[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[tt][ks], bf[ks], o[tt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);               // (the shortcut's conversions stay behind the MFMAs: so does their wait)
        const h8 s0 = __builtin_bit_cast(h8, cur[0]), s1 = __builtin_bit_cast(h8, cur[1]);
        h8 q0, q1;
        float qf[16];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int e = tt * 4 + j;
            float v = (o[tt][j] + b3v[e]) + (float)(e < 8 ? s0[e & 7] : s1[e & 7]);
            if (p.relu3) v = v < 0.0f ? 0.0f : v;
            qf[e] = v;
          }
        q0 = d_cvt8_f16<h8>(qf); q1 = d_cvt8_f16<h8>(qf + 8);
        const uint32_t off = m < M ? off0 + (uint32_t)pt * 16u * rowB : OOB;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, q0), ry3, (int)off, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, q1), ry3, (int)off, 64, 0);
      };
      int pt = 0;
      for (; pt + 4 <= NPT; pt += 4) {
        step(pt, r0, r2);
        step(pt + 1, r1, r3);
        step(pt + 2, r2, r0);
        step(pt + 3, r3, r1);
      }
      if (NPT % 4 == 2) {
        step(NPT - 2, r0, r2);
        step(NPT - 1, r1, r3);
      }
    }
    return;
  }
  // ---- epilogue: lane = pixel l15 of every pixel tile, channels 16 lq .. 16 lq + 15 of the wave's 64
  float bv[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) bv[e] = (float)bh[e >> 3][e & 7];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const long long m = tile_m * TM + wm * 16 * MT + mt * 16 + l15;
    if constexpr (TAPS == 9 && LIN) {
      if (p.pool) {
        // relu(conv + bias) of this lane's pixel (0 for a pixel of the even-rounded map that the real one does not have:
        // neutral under the maximum of ReLU outputs), then the maximum over the pooling window = the DPP quad; rounding to
        // float16 commutes with the maximum.  Lane 0 of the quad stores the pooled pixel m >> 2.
        const int Wp = (W + 1) & ~1, Hp = (H + 1) & ~1;
        const long long Sp = (long long)Hp * Wp;
        const long long img = m / Sp;
        const int rem = (int)(m - img * Sp);
        const int pair = rem / (2 * Wp), r2 = rem - pair * 2 * Wp;
        const bool real = m < M && (2 * pair + (r2 & 1)) < H && (r2 >> 1) < W;
        h8 o[2];
        float of[16];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float v = acc[mt][t][j] + bv[t * 4 + j];
            v = (v < 0.0f || !real) ? 0.0f : v;
            int vi = __builtin_bit_cast(int, v);
            float u = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, vi, 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
            v = u > v ? u : v;
            vi = __builtin_bit_cast(int, v);
            u = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, vi, 0x4E, 0xF, 0xF, false));         // quad_perm [2,3,0,1]
            v = u > v ? u : v;
            of[t * 4 + j] = v;
          }
        o[0] = d_cvt8_f16<h8>(of); o[1] = d_cvt8_f16<h8>(of + 8);
        if ((l15 & 3) == 0 && m < M) {
          _Float16* dst = p.y[lv] + (m >> 2) * cout + c0;
          *reinterpret_cast<h8*>(dst) = o[0];
          *reinterpret_cast<h8*>(dst + HOFF) = o[1];
        }
        continue;
      }
    }
    if (m < M) {
      h8 o[2];
      unsigned ou[8];
      if constexpr (TAPS == 1) {
        if (p.y32) {                                       // float32 out: the accumulators as they are, + float32 bias
          float* dst = p.y32 + m * cout + c0;
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            f4 v = acc[mt][t];
            if (p.bias32) v += *reinterpret_cast<const f4*>(p.bias32 + c0 + (t >> 1) * HOFF + 4 * (t & 1));
            if (p.relu) {
#pragma unroll
              for (int j = 0; j < 4; ++j) v[j] = v[j] < 0.0f ? 0.0f : v[j];
            }
            *reinterpret_cast<f4*>(dst + (t >> 1) * HOFF + 4 * (t & 1)) = v;
          }
          continue;
        }
        if (p.top) {
          // FPN top-down merge in the lateral convolution's epilogue (neck.hip's arithmetic: TF1 legacy resize, float32,
          // up * 0.5 + lateral * 0.5, one rounding) -- the lateral map never goes to memory
          const long long opx = (long long)p.Ho * p.Wo;
          const long long img = m / opx;
          const int rem = (int)(m - img * opx);
          const int yy = rem / p.Wo, xx = rem - yy * p.Wo;
          const float fy = (float)yy * p.tys, fx = (float)xx * p.txs;
          const float y0f = floorf(fy), x0f = floorf(fx);
          const int y0 = (int)y0f, x0 = (int)x0f;
          const int y1 = min(y0 + 1, p.th - 1), x1 = min(x0 + 1, p.tw - 1);
          const float yl = fy - y0f, xl = fx - x0f;
          const _Float16* tb = p.top + (img * p.th * p.tw) * cout + c0;
          h8 tl[2], tr[2], bl[2], br[2];
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            tl[hh] = *reinterpret_cast<const h8*>(tb + ((long long)y0 * p.tw + x0) * cout + HOFF * hh);
            tr[hh] = *reinterpret_cast<const h8*>(tb + ((long long)y0 * p.tw + x1) * cout + HOFF * hh);
            bl[hh] = *reinterpret_cast<const h8*>(tb + ((long long)y1 * p.tw + x0) * cout + HOFF * hh);
            br[hh] = *reinterpret_cast<const h8*>(tb + ((long long)y1 * p.tw + x1) * cout + HOFF * hh);
          }
#pragma unroll
          for (int e2 = 0; e2 < 8; ++e2) {
            float w2[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
              const int e = 2 * e2 + k, t = e >> 2, j = e & 3;
              const float lat = acc[mt][t][j] + bv[e];
              const float a = (float)tl[e >> 3][e & 7], b = (float)tr[e >> 3][e & 7];
              const float c = (float)bl[e >> 3][e & 7], d = (float)br[e >> 3][e & 7];
              const float tp = a + (b - a) * xl;
              const float bt = c + (d - c) * xl;
              const float up = tp + (bt - tp) * yl;
              w2[k] = up * 0.5f + lat * 0.5f;
            }
            // (the compiler's own conversions here: with the packed instruction as inline asm this path holds 28 more registers)
            typedef _Float16 h2v __attribute__((ext_vector_type(2)));
            const h2v q2 = {(_Float16)w2[0], (_Float16)w2[1]};
            ou[e2] = __builtin_bit_cast(unsigned, q2);
          }
          o[0] = d_pack8_f16<h8>(ou); o[1] = d_pack8_f16<h8>(ou + 4);
          _Float16* dst = p.y[lv] + m * cout + c0;
          *reinterpret_cast<h8*>(dst) = o[0];
          *reinterpret_cast<h8*>(dst + HOFF) = o[1];
          continue;
        }
      }
      h8 rs[2];
      const bool has_res = TAPS == 1 && p.res != nullptr;
      if (has_res) {
        rs[0] = *reinterpret_cast<const h8*>(p.res + m * cout + c0);
        rs[1] = *reinterpret_cast<const h8*>(p.res + m * cout + c0 + HOFF);
      }
#pragma unroll
      for (int e2 = 0; e2 < 8; ++e2) {
        float w2[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int e = 2 * e2 + k;
          float v = acc[mt][e >> 2][e & 3] + bv[e];
          if (has_res) v += (float)rs[e >> 3][e & 7];
          if (p.relu) v = v < 0.0f ? 0.0f : v;
          w2[k] = v;
        }
        ou[e2] = d_cvt_pk_f16(w2[0], w2[1]);
      }
      o[0] = d_pack8_f16<h8>(ou); o[1] = d_pack8_f16<h8>(ou + 4);
      _Float16* dst = p.y[lv] + m * cout + c0;
      *reinterpret_cast<h8*>(dst) = o[0];
      *reinterpret_cast<h8*>(dst + HOFF) = o[1];
    }
  }
  }   // tn_pass (one pass unless TAIL)
}

template <int MT, int WN, bool TAIL = false, bool BLK = false>
__global__ void __launch_bounds__(512) k_conv3x3_f16(Conv3x3Params p) {
  conv_tile_f16<MT, WN, TAIL, BLK, 9>(p);
}

// The pointwise form: 1x1 convolutions (optionally strided) and dense layers with K >= 128 as the same LDS-staged GEMM --
// the bottlenecks' first and shortcut convolutions (resnet_fpn.py:154-205), the neck's laterals with the top-down merge
// in the epilogue (resnet_fpn.py:339-398), the RoI head's dense layers (resnet_fpn.py:292-336).
template <int MT, int WN>
__global__ void __launch_bounds__(512) k_pointwise_f16(Conv3x3Params p) {
  conv_tile_f16<MT, WN, false, false, 1>(p);
}

// The RING forms (conv_tile_f16's header): 4 waves, 64 pixels x 64 channels, NS stages -- launches with few pixels.
template <int NS>
__global__ void __launch_bounds__(256) k_conv3x3_f16_ring(Conv3x3Params p) {
  conv_tile_f16<1, 1, false, false, 9, 4, NS>(p);
}
template <int NS>
__global__ void __launch_bounds__(256) k_pointwise_f16_ring(Conv3x3Params p) {
  conv_tile_f16<1, 1, false, false, 1, 4, NS>(p);
}

// (the float32 forms -- the detectors' parity mode -- live in conv_f32.hip)

// ---- host side: the tile table -------------------------------------------------------------------------------------------
// A tile = {waves, waves along the channels, 16-pixel tiles per wave, LDS stages}: TM = (nw / wn) * 16 * mt pixels x
// TN = 64 * wn channels.  ns == 2: the half-step-pipelined loop (launches that fill the chip); ns > 2: the ring forms.
struct ConvTile { int nw, wn, mt, ns; };
typedef void (*conv_kernel_t)(Conv3x3Params);
struct TileEntry { ConvTile t; conv_kernel_t plain, pw, blk; };
#define C3_LEGACY(MT_, WN_) \
  {{8, WN_, MT_, 2}, k_conv3x3_f16<MT_, WN_>, k_pointwise_f16<MT_, WN_>, k_conv3x3_f16<MT_, WN_, false, true>}
#define C3_RING(NS_) {{4, 1, 1, NS_}, k_conv3x3_f16_ring<NS_>, k_pointwise_f16_ring<NS_>, nullptr}
static const TileEntry kTiles[] = {
    C3_LEGACY(4, 4), C3_LEGACY(5, 4), C3_LEGACY(6, 4), C3_LEGACY(7, 4), C3_LEGACY(8, 4),       // 128 .. 256 px x 256 ch
    C3_LEGACY(2, 2), C3_LEGACY(3, 2), C3_LEGACY(4, 2),                                         // 128 .. 256 px x 128 ch
    C3_LEGACY(1, 1), C3_LEGACY(2, 1),                                                          // 128 / 256 px x 64 ch
    C3_LEGACY(2, 4), C3_LEGACY(3, 4), C3_LEGACY(1, 2),                                         // 64 / 96 px x 256 ch, 64 px x 128 ch
    // ring forms, 64 px x 64 ch: 8 stages (128 KB: one workgroup per CU, seven K-steps of copies in flight) and 4 stages
    // (two workgroups per CU).  Also built and measured in round 4, not kept (tools/r04/small_tiles.py, sweep of all layer
    // shapes at batch 1 / 2 / 4, cold L2): 64 x 128 (6 stages), 64 x 256 (3; with the fused tail), and rings for the 8-wave
    // 128 x 256 / 128 x 128 / 128 x 64 tiles -- within 3 % of the two-stage loop or behind it wherever they applied
    C3_RING(8), C3_RING(4),
};
#undef C3_LEGACY
#undef C3_RING
static constexpr int kNumTiles = (int)(sizeof(kTiles) / sizeof(kTiles[0]));

static const TileEntry* find_tile(const ConvTile& t) {
  for (const TileEntry& e : kTiles)
    if (e.t.nw == t.nw && e.t.wn == t.wn && e.t.mt == t.mt && e.t.ns == t.ns) return &e;
  return nullptr;
}
static inline int tile_tm(const ConvTile& t) { return (t.nw / t.wn) * 16 * t.mt; }
static inline unsigned tile_lds(const ConvTile& t) { return (unsigned)t.ns * (unsigned)(tile_tm(t) + 64 * t.wn) * 128u; }

static hipError_t tiles_init() {                // (more than the default 64 KB of dynamic LDS)
  static OdetPerDeviceOnce once;
  return once.run([] {
    hipError_t rc = hipSuccess;
    auto set = [&rc](const void* k_) {
      const hipError_t e_ = hipFuncSetAttribute(k_, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e_ != hipSuccess) rc = e_;
    };
    for (const TileEntry& e : kTiles) {
      set((const void*)e.plain); set((const void*)e.pw);
      if (e.blk) set((const void*)e.blk);
    }
    set((const void*)k_conv3x3_f16<4, 4, true>); set((const void*)k_conv3x3_f16<5, 4, true>);
    set((const void*)k_conv3x3_f16<6, 4, true>); set((const void*)k_conv3x3_f16<7, 4, true>);
    set((const void*)k_conv3x3_f16<8, 4, true>);
    return rc;
  });
}

static int launch_tile(conv_kernel_t k, const ConvTile& t, unsigned blocks, unsigned lds_bytes, Conv3x3Params& p, hipStream_t st) {
  void* args[1] = {&p};
  ODET_HIP(hipLaunchKernel((const void*)k, dim3(blocks), dim3((unsigned)t.nw * 64u), args, lds_bytes, st));
  return ODET_OK;
}

// ---- tile selection for launches that do not fill the chip ---------------------------------------------------------------
// The 8-wave tiles (128 .. 256 pixels) are picked by the measured rules inside the launchers below.  When a layer has so few
// pixels that 64 x 64 tiles still make no more workgroups than two per CU -- batch 1 .. 2 on the 50 x 84 / 25 x 42 maps, the
// RoI head's dense layers at 1000 rows: the BASELINE configs' own batch size -- it goes to the ring form: every CU gets
// work, and the copies of several K-steps are in flight instead of one (a launch's operands sit in the Infinity Cache,
// not in the XCD's L2 -- the producer ran on other XCDs --, and with two stages every K-step waited for that round trip:
// conv4's first 1x1 at batch 1 took 22 us inside a pass, 16 K-steps of 1.3 us).  Measured with cold L2s
// (tools/r04/small_tiles.py, profiles/r04_small_tiles.json): conv4's 3x3 at batch 1 18.8 us (two-stage 128 x 64 tiles 24.8,
// library 25.1), its first 1x1 9.6 (12.7 / 11.7), conv5's 3x3 at batch 2 31.6 (39.9 / 31.2).  What bounds these launches is
// a CU's intake from L2 (~30 B / clock with enough copies in flight), so a 64 x 64 x 64 K-step takes ~0.27 us whatever the
// matrix pipe could do; beyond two workgroups per CU the larger tiles' better ratio of matrix work to copied bytes wins.
// A pick that makes fewer workgroups than the chip has CUs (a "sub-round" launch: batch 2 .. 8 on the 50 x 84 maps, batch
// 1 on the 100 x 167 ones): the 128 x 128 and 128 x 64 tiles make two or four times as many, and their 64 / 48 KB of
// stages let two share a CU.  Priced with the launchers' own per-K-step model plus what sharing a CU costs (measured,
// tools/r04/small_tiles.py, cold L2: conv4's 3x3 at batch 4 33.0 vs 37.6 us, its first 1x1 16.8 vs 20.3, its last 1x1 at
// batch 1 10.2 vs 12.1, the neck's 100 x 167 smoothing convolution at batch 1 32.0 vs 38.0).
static ConvTile refine_sub_round(const long long* level_px, int num_levels, int cout, int ksteps, ConvTile pick) {
  auto blocks_of = [&](const ConvTile& t) {
    long long slabs = 0;
    for (int l = 0; l < num_levels; ++l) slabs += (level_px[l] + tile_tm(t) - 1) / tile_tm(t);
    return (slabs + 7) / 8 * 8 * (long long)(cout / (64 * t.wn));
  };
  if (blocks_of(pick) >= 256) return pick;
  auto cost_of = [&](const ConvTile& t) {
    const int tm = tile_tm(t), tn = 64 * t.wn;
    const long long blocks = blocks_of(t);
    const int occ = tile_lds(t) <= 80u * 1024u ? 2 : 1;
    const long long rounds = (blocks + 256ll * occ - 1) / (256ll * occ);
    const double share = (double)std::min<long long>(occ, (blocks + 255) / 256);
    return (double)rounds * (ksteps * std::max(tm * tn / 32.0, 2.0 * (tm + tn)) + 150.0) * (1.0 + 0.3 * (share - 1.0));
  };
  double best = cost_of(pick);
  const ConvTile cands[2] = {{8, 2, 2, 2}, {8, 1, 1, 2}};
  for (const ConvTile& c : cands) {
    if (cout % (64 * c.wn)) continue;
    const double cost = cost_of(c);
    if (cost < best * 0.95) { best = cost; pick = c; }
  }
  return pick;
}

static ConvTile pick_small(const long long* level_px, int num_levels, int cout, ConvTile pick) {
  long long slabs64 = 0;
  for (int l = 0; l < num_levels; ++l) slabs64 += (level_px[l] + 63) / 64;
  const long long n64 = slabs64 * (cout / 64);
  if (n64 <= 256) return ConvTile{4, 1, 1, 8};
  if (n64 <= 512) return ConvTile{4, 1, 1, 4};
  return pick;
}

#ifdef ODET_DIAG
// Diagnostic build only (-DODET_DIAG: tools/libodet_hip_diag.so, include/odet_diag.h; the shipped library has neither the entry
// point nor the override): force the tile of the 3x3 (form 0; the fused tail included) / pointwise (form 1) launches of this
// process; nw = 0 clears.
static std::atomic<unsigned> g_tile_override[2] = {{0u}, {0u}};
extern "C" int odet_debug_conv_tile(int form, int nw, int wn, int mt, int ns) {
  ODET_REQUIRE(form == 0 || form == 1, "odet_debug_conv_tile: form 0 (3x3) or 1 (pointwise)");
  if (nw == 0) { g_tile_override[form].store(0u); return ODET_OK; }
  ODET_REQUIRE(find_tile(ConvTile{nw, wn, mt, ns}) != nullptr, "odet_debug_conv_tile: no such tile (nw %d, wn %d, mt %d, ns %d)",
               nw, wn, mt, ns);
  g_tile_override[form].store((unsigned)nw << 24 | (unsigned)wn << 16 | (unsigned)mt << 8 | (unsigned)ns);
  return ODET_OK;
}
static bool tile_override(int form, int cout, int need_wn, ConvTile* t) {
  const unsigned v = g_tile_override[form].load();
  if (!v) return false;
  const ConvTile o{(int)(v >> 24), (int)(v >> 16 & 255), (int)(v >> 8 & 255), (int)(v & 255)};
  if (cout % (64 * o.wn) || (need_wn && o.wn != need_wn)) return false;
  *t = o;
  return true;
}
#else
static inline bool tile_override(int, int, int, ConvTile*) { return false; }   // (the shipped library: no process-global override)
#endif

struct Conv3x3Tail {             // the fused RpnHead tail (nullable in conv3x3_launch)
  const void* w; const void* b; int A; float* scores; long long s_stride; float* deltas; long long d_stride;
};
struct Conv3x3Block {            // the fused bottleneck tail (nullable in conv3x3_launch)
  const void* w3; const void* b3; const void* res; void* y3; int n3; int relu3;
};

static int conv3x3_launch(const odet_conv_level_t* levels, int num_levels, const void* w, const void* bias, int batch,
                          int cin, int cout, int relu, hipStream_t st, const Conv3x3Tail* tail = nullptr,
                          const Conv3x3Block* blk = nullptr, int pool = 0) {
  ODET_REQUIRE(levels && w, "odet_conv3x3_f16: null pointer");
  ODET_REQUIRE(num_levels >= 1 && num_levels <= ODET_MAX_LEVELS, "odet_conv3x3_f16: num_levels %d out of range", num_levels);
  ODET_REQUIRE(batch > 0, "odet_conv3x3_f16: bad batch");
  ODET_REQUIRE(cin > 0 && cin % C3_BK == 0, "odet_conv3x3_f16: cin %d must be a multiple of %d", cin, C3_BK);
  ODET_REQUIRE(cout > 0 && cout % 64 == 0, "odet_conv3x3_f16: cout %d must be a multiple of 64", cout);
  ODET_REQUIRE((unsigned long long)cout * 9ull * cin * 2ull < 0x7FFFFFFFull, "odet_conv3x3_f16: weights too large");
  ODET_REQUIRE(((uintptr_t)w | (uintptr_t)bias) % 16 == 0, "odet_conv3x3_f16: weights and bias must be 16-byte aligned");
  ODET_HIP(tiles_init());
  Conv3x3Params p;
  long long total = 0;
  for (int l = 0; l < ODET_MAX_LEVELS; ++l) {
    const odet_conv_level_t& L = levels[l < num_levels ? l : 0];
    ODET_REQUIRE(L.x && (L.y || tail || blk) && L.H > 0 && L.W > 0, "odet_conv3x3_f16: bad level %d", l);
    const long long Mreal = (long long)batch * L.H * L.W;
    // (pooled form: the launch's index space is the map rounded up to even sizes)
    const long long M = pool ? (long long)batch * ((L.H + 1) & ~1) * ((L.W + 1) & ~1) : Mreal;
    // 32-bit byte offsets into x (+ the padding rows of the descriptor) and the out-of-range marker
    ODET_REQUIRE((unsigned long long)Mreal * cin * 2ull + 2ull * (L.W + 1) * cin * 2ull < 0xFFFFFFF0ull,
                 "odet_conv3x3_f16: level %d input larger than 4 GiB", l);
    p.x[l] = (const _Float16*)L.x; p.y[l] = (_Float16*)L.y; p.M[l] = M; p.H[l] = L.H; p.W[l] = L.W;
    p.px[l] = (long long)L.H * L.W;
  }
  p.tail_w = nullptr; p.tail_b = nullptr; p.scores = nullptr; p.deltas = nullptr; p.s_stride = p.d_stride = 0; p.A = 0;
  p.w3 = nullptr; p.b3 = nullptr; p.res = nullptr; p.y3 = nullptr; p.n3 = 0; p.relu3 = 0;
  p.stride = 1; p.Ho = p.Wo = 0; p.Min = 0; p.top = nullptr; p.th = p.tw = 0; p.tys = p.txs = 0.0f;
  p.y32 = nullptr; p.bias32 = nullptr; p.x2 = nullptr; p.cin2 = 0; p.k1steps = 0; p.Min2 = 0;
  p.pool = pool ? 1 : 0;
  if (pool) {
    ODET_REQUIRE(num_levels == 1 && !tail && !blk && relu, "odet_conv3x3_relu_pool2_f16: one map, plain form, with ReLU");
    p.Min = (long long)batch * levels[0].H * levels[0].W;
  }
  if (blk) {
    ODET_REQUIRE(blk->w3 && blk->b3 && blk->y3 && bias, "odet_conv3x3_conv1x1_f16: null pointer");
    ODET_REQUIRE(cout == 256 || cout == 128 || cout == 64,
                 "odet_conv3x3_conv1x1_f16: the 3x3 convolution must have 64, 128 or 256 output channels (got %d)", cout);
    ODET_REQUIRE(blk->n3 > 0 && blk->n3 % 64 == 0, "odet_conv3x3_conv1x1_f16: n3 %d must be a multiple of 64", blk->n3);
    ODET_REQUIRE(num_levels == 1, "odet_conv3x3_conv1x1_f16: one map");
    // (the last convolution's shortcut and output rows are addressed through 32-bit buffer offsets)
    ODET_REQUIRE((unsigned long long)batch * levels[0].H * levels[0].W * (unsigned long long)blk->n3 * 2ull <= 0xFFFFFFF0ull,
                 "odet_conv3x3_conv1x1_f16: the output map must be smaller than 4 GiB");
    ODET_REQUIRE(((uintptr_t)blk->w3 | (uintptr_t)blk->res | (uintptr_t)blk->y3) % 16 == 0, "odet_conv3x3_conv1x1_f16: pointers must be 16-byte aligned");
    p.w3 = (const _Float16*)blk->w3; p.b3 = (const _Float16*)blk->b3; p.res = (const _Float16*)blk->res;
    p.y3 = (_Float16*)blk->y3; p.n3 = blk->n3; p.relu3 = blk->relu3 ? 1 : 0;
  }
  if (tail) {
    ODET_REQUIRE(tail->w && tail->b && tail->scores && tail->deltas && bias, "odet_rpn_head_fused_f16: null pointer");
    ODET_REQUIRE(tail->A >= 1 && 6 * tail->A <= 32, "odet_rpn_head_fused_f16: 1 <= A <= 5");
    ODET_REQUIRE(cout == 256 || cout == 512, "odet_rpn_head_fused_f16: cout must be 256 or 512 (got %d)", cout);
    long long a0 = 0;
    for (int l = 0; l < ODET_MAX_LEVELS; ++l) {
      p.aoff[l] = a0;
      if (l < num_levels) a0 += p.px[l] * tail->A;
    }
    ODET_REQUIRE(a0 * 2 <= tail->s_stride && a0 * 4 <= tail->d_stride, "odet_rpn_head_fused_f16: the levels do not fit the arrays");
    p.tail_w = (const _Float16*)tail->w; p.tail_b = (const _Float16*)tail->b; p.A = tail->A;
    p.scores = tail->scores; p.deltas = tail->deltas; p.s_stride = tail->s_stride; p.d_stride = tail->d_stride;
  }
  // Channel tile: 256 (four waves along the channels), or 128 / 64 for the layers with fewer output channels.
  const int wn_sel = (cout % 256 == 0) ? 4 : (cout % 128 == 0 ? 2 : 1);
  const int wm_sel = 8 / wn_sel;
  // Pixel-tile height: the launch runs in rounds of 256 workgroups (one per CU: 128 KB of LDS each), so a layer whose
  // 256-pixel slabs fill a round badly (ResNet's conv4 at batch 8: 132 slabs) is cut into 128 .. 224-pixel slabs
  // instead.  Cost model: rounds x (pixel tiles + 2) (a workgroup's time is its pixel tiles + the weight traffic they
  // share).
  const int mt_hi = 16 / wm_sel, mt_lo = 8 / wm_sel;          // TM = wm * 16 * mt in 128 .. 256
  int mt_best = mt_hi;
  {
    double best = 1e300;
    for (int mt = mt_hi; mt >= mt_lo; --mt) {
      const int tm = wm_sel * 16 * mt;
      long long slabs = 0;
      for (int l = 0; l < num_levels; ++l) slabs += (p.M[l] + tm - 1) / tm;
      // (fused RpnHead: a workgroup walks the channel tiles of its slab itself)
      const int tiles_mt = cout / (64 * wn_sel);
      const long long blocks_mt = (slabs + 7) / 8 * 8 * (tail ? 1 : tiles_mt);
      // plain launches ask for two stages of their own tile: tiles of <= 80 KB run two workgroups per CU (each at ~1 / 1.7
      // of the speed it has alone: they overlap each other's staging, barriers and epilogues)
      const int occ = (!tail && !blk && 2 * (tm + 64 * wn_sel) * 128 <= 80 * 1024) ? 2 : 1;
      const double cost = (double)((blocks_mt + 256 * occ - 1) / (256 * occ)) * (tm / 32 + 2) * (occ == 2 ? 1.7 : 1.0) * (tail ? tiles_mt : 1);
      if (cost < best * 0.97) { best = cost; mt_best = mt; }       // (smaller tiles only for a clear gain)
    }
  }
  ConvTile tile{8, wn_sel, mt_best, 2};
  if (!tail && !pool) {
    // few pixels (conv4 / conv5 / the small neck levels at batch 1 .. 2): the ring form
    if (!blk) tile = pick_small(p.M, num_levels, cout, refine_sub_round(p.M, num_levels, cout, 9 * (cin / C3_BK), tile));
    tile_override(0, cout, blk ? wn_sel : 0, &tile);
    ODET_REQUIRE(!blk || tile.ns == 2, "odet_bottleneck_tail_f16: the fused tail has no ring form");
  }
  const TileEntry* te = find_tile(tile);
  ODET_REQUIRE(te != nullptr, "odet_conv3x3_f16: internal: no kernel for the picked tile");
  p.tiles_n = cout / (64 * tile.wn);
  const int TMsel = tile_tm(tile);
  for (int l = 0; l < ODET_MAX_LEVELS; ++l) {
    p.tile_start[l] = total;
    if (l < num_levels) total += (p.M[l] + TMsel - 1) / TMsel;
  }
  p.tile_start[ODET_MAX_LEVELS] = total;
  for (int l = num_levels; l <= ODET_MAX_LEVELS; ++l) p.tile_start[l] = total;
  p.w = (const _Float16*)w; p.bias = (const _Float16*)bias;
  p.num_levels = num_levels; p.cin = cin; p.cout = cout; p.relu = relu ? 1 : 0;
  const long long groups = (total + 7) / 8;
  const long long blocks = groups * 8 * (tail ? 1 : p.tiles_n);
  ODET_REQUIRE(blocks < (1ll << 31), "odet_conv3x3_f16: too many workgroups");
  if (tail) {
    conv_kernel_t kt = k_conv3x3_f16<8, 4, true>;
    switch (mt_best) {
      case 4: kt = k_conv3x3_f16<4, 4, true>; break;
      case 5: kt = k_conv3x3_f16<5, 4, true>; break;
      case 6: kt = k_conv3x3_f16<6, 4, true>; break;
      case 7: kt = k_conv3x3_f16<7, 4, true>; break;
      default: break;
    }
    // (the K loop's stages + 128 bytes per pixel for the sums over the channel tiles)
    return launch_tile(kt, tile, (unsigned)blocks, C3_LDS_BYTES + (unsigned)TMsel * 128u, p, st);
  }
  // LDS: the K loop's stages of the launch's own tile; the fused tail re-uses them for the TM x 2 CMID-byte activation tile
  unsigned lds_bytes = tile_lds(tile);
  if (blk) lds_bytes = std::max(lds_bytes, (unsigned)TMsel * 128u * (unsigned)tile.wn);
  return launch_tile(blk ? te->blk : te->plain, tile, (unsigned)blocks, lds_bytes, p, st);
}

extern "C" int odet_conv3x3_f16(const void* x, const void* w, const void* bias, void* y, int batch, int H, int W, int cin,
                                int cout, int relu, odet_stream_t stream) {
  ODET_REQUIRE(x && y, "odet_conv3x3_f16: null pointer");
  const odet_conv_level_t one{x, y, H, W};
  return conv3x3_launch(&one, 1, w, bias, batch, cin, cout, relu, (hipStream_t)stream);
}

extern "C" int odet_conv3x3_relu_pool2_f16(const void* x, const void* w, const void* bias, void* y, int batch, int H, int W,
                                           int cin, int cout, odet_stream_t stream) {
  ODET_REQUIRE(x && y, "odet_conv3x3_relu_pool2_f16: null pointer");
  const odet_conv_level_t one{x, y, H, W};
  return conv3x3_launch(&one, 1, w, bias, batch, cin, cout, 1, (hipStream_t)stream, nullptr, nullptr, 1);
}

extern "C" int odet_conv3x3_f16_levels(const odet_conv_level_t* levels, int num_levels, const void* w, const void* bias,
                                       int batch, int cin, int cout, int relu, odet_stream_t stream) {
  return conv3x3_launch(levels, num_levels, w, bias, batch, cin, cout, relu, (hipStream_t)stream);
}

extern "C" int odet_rpn_head_fused_f16(const odet_conv_level_t* levels, int num_levels, const void* conv_w, const void* conv_b,
                                       const void* w, const void* b, int A, int batch, int cin, int cout, float* scores,
                                       long long scores_image_stride, float* deltas, long long deltas_image_stride,
                                       odet_stream_t stream) {
  ODET_REQUIRE(((uintptr_t)w | (uintptr_t)conv_b) % 16 == 0, "odet_rpn_head_fused_f16: pointers must be 16-byte aligned");
  const Conv3x3Tail t{w, b, A, scores, scores_image_stride, deltas, deltas_image_stride};
  return conv3x3_launch(levels, num_levels, conv_w, conv_b, batch, cin, cout, 1, (hipStream_t)stream, &t);
}

extern "C" int odet_conv3x3_conv1x1_f16(const void* x, const void* w2, const void* b2, const void* w3, const void* b3,
                                        const void* residual, void* y, int batch, int H, int W, int cin, int n3, int relu,
                                        odet_stream_t stream) {
  ODET_REQUIRE(x && y, "odet_conv3x3_conv1x1_f16: null pointer");
  const odet_conv_level_t one{x, nullptr, H, W};
  const Conv3x3Block b{w3, b3, residual, y, n3, relu};
  return conv3x3_launch(&one, 1, w2, b2, batch, cin, 256, 1, (hipStream_t)stream, nullptr, &b);
}

// ---- pointwise form: host side ------------------------------------------------------------------------------------------
struct PwEpilogue { const void* res; const void* top; int th, tw; float* y32; const float* bias32;
                    const void* x2; int cin2; };     // x2: second source along K ([batch][H][W][cin2], strided; x is then [M][cin])

static int pointwise_launch(const char* who, const void* x, const void* w, const void* bias, void* y, int batch, int H, int W,
                            int stride, int cin, int cout, int relu, const PwEpilogue& epi, hipStream_t st) {
  ODET_REQUIRE(x && w && (y || epi.y32), "%s: null pointer", who);
  ODET_REQUIRE(batch > 0 && H > 0 && W > 0 && (stride == 1 || stride == 2), "%s: bad shape", who);
  ODET_REQUIRE(!epi.y32 || (!epi.res && !epi.top && !bias && ((uintptr_t)epi.y32 | (uintptr_t)epi.bias32) % 16 == 0),
               "%s: bad float32-output arguments", who);
  ODET_REQUIRE(cin % C3_BK == 0 && cin + (epi.x2 ? epi.cin2 : 0) >= 2 * C3_BK,
               "%s: cin %d must be a multiple of %d, at least %d along K", who, cin, C3_BK, 2 * C3_BK);
  ODET_REQUIRE(cout > 0 && cout % 64 == 0, "%s: cout %d must be a multiple of 64", who, cout);
  ODET_REQUIRE((unsigned long long)cout * cin * 2ull < 0x7FFFFFFFull, "%s: weights too large", who);
  ODET_REQUIRE(((uintptr_t)x | (uintptr_t)w | (uintptr_t)y | (uintptr_t)bias | (uintptr_t)epi.res | (uintptr_t)epi.top) % 16 == 0,
               "%s: pointers must be 16-byte aligned", who);
  ODET_REQUIRE(!(epi.res && epi.top), "%s: shortcut and top-down merge exclude each other", who);
  ODET_REQUIRE(!epi.top || (stride == 1 && epi.th > 0 && epi.tw > 0 && !relu), "%s: bad merge arguments", who);
  ODET_HIP(tiles_init());
  const int Ho = (H + stride - 1) / stride, Wo = (W + stride - 1) / stride;
  const long long M = (long long)batch * Ho * Wo;
  const long long Min = epi.x2 ? M : (long long)batch * H * W;       // (two sources: x has the OUTPUT's rows)
  ODET_REQUIRE((unsigned long long)Min * cin * 2ull < 0xFFFFFFF0ull, "%s: input larger than 4 GiB", who);
  ODET_REQUIRE(!epi.x2 || (epi.cin2 > 0 && epi.cin2 % C3_BK == 0 && (uintptr_t)epi.x2 % 16 == 0 &&
                           (unsigned long long)batch * H * W * epi.cin2 * 2ull < 0xFFFFFFF0ull &&
                           (unsigned long long)cout * (cin + epi.cin2) * 2ull < 0x7FFFFFFFull), "%s: bad second source", who);
  Conv3x3Params p;
  for (int l = 0; l < ODET_MAX_LEVELS; ++l) {
    p.x[l] = (const _Float16*)x; p.y[l] = (_Float16*)y; p.M[l] = M; p.H[l] = H; p.W[l] = W; p.px[l] = (long long)Ho * Wo;
    p.aoff[l] = 0;
  }
  p.tail_w = nullptr; p.tail_b = nullptr; p.scores = nullptr; p.deltas = nullptr; p.s_stride = p.d_stride = 0; p.A = 0;
  p.w3 = nullptr; p.b3 = nullptr; p.y3 = nullptr; p.n3 = 0; p.relu3 = 0;
  p.res = (const _Float16*)epi.res;
  p.top = (const _Float16*)epi.top; p.th = epi.th; p.tw = epi.tw;
  p.tys = epi.top ? (float)epi.th / (float)Ho : 0.0f;
  p.txs = epi.top ? (float)epi.tw / (float)Wo : 0.0f;
  p.stride = stride; p.Ho = Ho; p.Wo = Wo; p.Min = Min;
  p.y32 = epi.y32; p.bias32 = epi.bias32;
  p.x2 = (const _Float16*)epi.x2; p.cin2 = epi.x2 ? epi.cin2 : 0; p.k1steps = cin / C3_BK; p.Min2 = (long long)batch * H * W;
  p.pool = 0;
  // Tile: channels 256 / 128 / 64 (WN = 4 / 2 / 1 waves along the channels) x pixels (8 / WN) * 16 * MT.  One workgroup
  // per CU (128 KB of LDS), so the launch runs in rounds of 256 workgroups; per K-step a workgroup needs about
  // max(matrix cycles TM * TN / 32, staging cycles 2 * (TM + TN)).  Pick the pair with the least rounds x that.
  int wn_best = 0, mt_best = 0;
  double best = 1e300;
  for (int wn = 4; wn >= 1; wn >>= 1) {
    if (cout % (64 * wn)) continue;
    const int wm = 8 / wn, tn = 64 * wn, tiles_n = cout / tn;
    for (int mt = 16 / wm; mt >= 8 / wm; --mt) {
      const int tm = wm * 16 * mt;
      const long long slabs = (M + tm - 1) / tm;
      const long long blocks = (slabs + 7) / 8 * 8 * tiles_n;
      const double per = std::max((double)tm * tn / 32.0, 2.0 * (tm + tn)) + 150.0;
      const double cost = (double)((blocks + 255) / 256) * per;
      if (cost < best * 0.97) { best = cost; wn_best = wn; mt_best = mt; }
    }
  }
  ConvTile tile{8, wn_best, mt_best, 2};
  // fewer workgroups than the chip holds (few pixels: batch 1 .. 4 on the small maps, the RoI head's dense layers at 1000
  // rows): the ring forms
  tile = pick_small(&M, 1, cout, refine_sub_round(&M, 1, cout, (cin + (epi.x2 ? epi.cin2 : 0)) / C3_BK, tile));
  tile_override(1, cout, 0, &tile);
  const TileEntry* te = find_tile(tile);
  ODET_REQUIRE(te != nullptr, "%s: internal: no kernel for the picked tile", who);
  const int TMsel = tile_tm(tile);
  p.tiles_n = cout / (64 * tile.wn);
  const long long total = (M + TMsel - 1) / TMsel;
  p.tile_start[0] = 0;
  for (int l = 1; l <= ODET_MAX_LEVELS; ++l) p.tile_start[l] = total;
  p.w = (const _Float16*)w; p.bias = (const _Float16*)bias;
  p.num_levels = 1; p.cin = cin; p.cout = cout; p.relu = relu ? 1 : 0;
  const long long blocks = (total + 7) / 8 * 8 * p.tiles_n;
  ODET_REQUIRE(blocks < (1ll << 31), "%s: too many workgroups", who);
  return launch_tile(te->pw, tile, (unsigned)blocks, tile_lds(tile), p, st);
}

extern "C" int odet_pointwise_f16(const void* x, const void* w, const void* bias, const void* residual, void* y, int batch,
                                  int H, int W, int stride, int cin, int cout, int relu, odet_stream_t stream) {
  const PwEpilogue e{residual, nullptr, 0, 0, nullptr, nullptr, nullptr, 0};
  return pointwise_launch("odet_pointwise_f16", x, w, bias, y, batch, H, W, stride, cin, cout, relu, e, (hipStream_t)stream);
}

extern "C" int odet_lateral_merge_f16(const void* x, const void* w, const void* bias, const void* top, int th, int tw, void* y,
                                      int batch, int H, int W, int cin, int cout, odet_stream_t stream) {
  ODET_REQUIRE(top, "odet_lateral_merge_f16: null pointer");
  const PwEpilogue e{nullptr, top, th, tw, nullptr, nullptr, nullptr, 0};
  return pointwise_launch("odet_lateral_merge_f16", x, w, bias, y, batch, H, W, 1, cin, cout, 0, e, (hipStream_t)stream);
}

extern "C" int odet_pointwise_dual_f16(const void* x1, int cin1, const void* x2, int cin2, int H2, int W2, int stride2,
                                       const void* w, const void* bias, void* y, int batch, int cout, int relu,
                                       odet_stream_t stream) {
  ODET_REQUIRE(x2, "odet_pointwise_dual_f16: null pointer");
  const PwEpilogue e{nullptr, nullptr, 0, 0, nullptr, nullptr, x2, cin2};
  return pointwise_launch("odet_pointwise_dual_f16", x1, w, bias, y, batch, H2, W2, stride2, cin1, cout, relu, e,
                          (hipStream_t)stream);
}

extern "C" int odet_dense_f16_out_f32(const void* x, const void* w, const float* bias, float* y, long long rows, int cin,
                                      int cout, int relu, odet_stream_t stream) {
  ODET_REQUIRE(y && rows > 0 && rows < (1ll << 31), "odet_dense_f16_out_f32: bad arguments");
  const PwEpilogue e{nullptr, nullptr, 0, 0, y, bias, nullptr, 0};
  return pointwise_launch("odet_dense_f16_out_f32", x, w, nullptr, nullptr, 1, 1, (int)rows, 1, cin, cout, relu, e,
                          (hipStream_t)stream);
}

extern "C" int odet_bottleneck_tail_f16(const void* x, const void* w2, const void* b2, const void* w3, const void* b3,
                                       const void* residual, void* y, int batch, int H, int W, int cin, int cmid, int n3,
                                       int relu, odet_stream_t stream) {
  ODET_REQUIRE(x && y, "odet_bottleneck_tail_f16: null pointer");
  const odet_conv_level_t one{x, nullptr, H, W};
  const Conv3x3Block b{w3, b3, residual, y, n3, relu};
  return conv3x3_launch(&one, 1, w2, b2, batch, cin, cmid, 1, (hipStream_t)stream, nullptr, &b);
}


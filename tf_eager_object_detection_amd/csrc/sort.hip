// Stable LSD radix sort of (score, index) pairs, descending score / ascending index on ties.
//
// Feeds the exact greedy NMS (model/region_proposal.py:74 runs tf.image.non_max_suppression
// over ALL anchors: 267 069 at 800x1333 FPN).  TF pops a max-heap; a stable descending sort
// gives the same visiting order with the declared tie rule (score desc, index asc).
//
// This is the FALLBACK ordering of the NMS path: the common case orders only the best ~1.5 K
// candidates (radix select + one-workgroup sort, nms.hip); the full sort runs when that chunk does
// not reach max_output.  Every kernel takes a `skip` flag (the NMS "done" word in device memory) and
// exits at once when it is set, so the sync-free mode can enqueue the fallback unconditionally.
//
// Key: ~asc(score) so that an ascending LSD sort yields descending scores (keys are produced by
// k_rp_prepare in nms.hip; scores TF would never push -- NaN or <= lowest float -- have key
// 0xFFFFFFFF and sort to the end).
//
// 4 passes x 8-bit digits, 2 launches per pass:
//   k_rs_hist    : per-block digit histogram, stored block-major (one coalesced 1 KiB row per block)
//   k_rs_scatter : every block derives its own global offsets from the histogram table (256 threads =
//                  256 digits, column sums over the L2-resident table + one block scan -- no separate
//                  scan launch), ranks its 2048 keys with a wave64 ballot multi-split (8 ballots give
//                  each lane the set of lanes sharing its digit) and scatters keys + payload.
// Pass 0's histogram is produced by k_rs_init (which also writes the iota payload).
#include "odet_internal.h"

#define RS_BLOCK 256
#define RS_ITEMS 8
#define RS_TILE (RS_BLOCK * RS_ITEMS)
#define RS_WAVES (RS_BLOCK / 64)
#define RS_RADIX 256

size_t odet_sort_hist_entries(int n) {
  size_t nblocks = ((size_t)(n > 0 ? n : 1) + RS_TILE - 1) / RS_TILE;
  return nblocks * RS_RADIX;
}

// element order inside a block: wave-contiguous: e = blockStart + (wave*ITEMS + item)*64 + lane
__device__ __forceinline__ int rs_elem(int item) {
  int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  return blockIdx.x * RS_TILE + (w * RS_ITEMS + item) * 64 + lane;
}

// One image of a batched sort (blockIdx.y): its two key / payload buffers, its histogram table and its `skip` word.
// `flip` = 0: pass reads (ka, va) and writes (kb, vb); 1: the other way round.
struct RsImg { uint32_t* ka; uint32_t* va; uint32_t* kb; uint32_t* vb; uint32_t* hist; const int32_t* skip; };
struct RsBatch { RsImg v[ODET_MAX_BATCH]; };

// payload (iota) + pass-0 histogram from precomputed keys
__global__ void __launch_bounds__(RS_BLOCK) k_rs_init(RsBatch bt, int n) {
  const RsImg& im = bt.v[blockIdx.y];
  const int32_t* __restrict__ skip = im.skip;
  if (skip && *skip) return;
  const uint32_t* __restrict__ keys = im.ka;
  uint32_t* __restrict__ vals = im.va;
  uint32_t* __restrict__ hist = im.hist;
  __shared__ uint32_t h[RS_RADIX];
  h[threadIdx.x] = 0;
  __syncthreads();
#pragma unroll
  for (int it = 0; it < RS_ITEMS; ++it) {
    int e = rs_elem(it);
    if (e < n) {
      vals[e] = (uint32_t)e;
      atomicAdd(&h[keys[e] & 0xFF], 1u);
    }
  }
  __syncthreads();
  hist[(size_t)blockIdx.x * RS_RADIX + threadIdx.x] = h[threadIdx.x];   // block-major
}

__global__ void __launch_bounds__(RS_BLOCK) k_rs_hist(RsBatch bt, int n, int shift, int flip) {
  const RsImg& im = bt.v[blockIdx.y];
  const int32_t* __restrict__ skip = im.skip;
  if (skip && *skip) return;
  const uint32_t* __restrict__ keys = flip ? im.kb : im.ka;
  uint32_t* __restrict__ hist = im.hist;
  __shared__ uint32_t h[RS_RADIX];
  h[threadIdx.x] = 0;
  __syncthreads();
#pragma unroll
  for (int it = 0; it < RS_ITEMS; ++it) {
    int e = rs_elem(it);
    if (e < n) atomicAdd(&h[(keys[e] >> shift) & 0xFF], 1u);
  }
  __syncthreads();
  hist[(size_t)blockIdx.x * RS_RADIX + threadIdx.x] = h[threadIdx.x];
}

__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v) {
  int lane = threadIdx.x & 63;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t t = __shfl_up(v, d);
    if (lane >= d) v += t;
  }
  return v;
}

__global__ void __launch_bounds__(RS_BLOCK) k_rs_scatter(RsBatch bt, int n, int shift, int nblocks, int flip) {
  const RsImg& im = bt.v[blockIdx.y];
  const int32_t* __restrict__ skip = im.skip;
  if (skip && *skip) return;
  const uint32_t* __restrict__ keys_in = flip ? im.kb : im.ka;
  const uint32_t* __restrict__ vals_in = flip ? im.vb : im.va;
  uint32_t* __restrict__ keys_out = flip ? im.ka : im.kb;
  uint32_t* __restrict__ vals_out = flip ? im.va : im.vb;
  const uint32_t* __restrict__ hist = im.hist;
  __shared__ uint32_t cnt[RS_WAVES][RS_RADIX];
  __shared__ uint32_t gbase[RS_RADIX];
  __shared__ uint32_t wsum[RS_WAVES];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int k = threadIdx.x; k < RS_WAVES * RS_RADIX; k += RS_BLOCK) (&cnt[0][0])[k] = 0;

  // issue this block's key loads first so they overlap the offset computation
  uint32_t key[RS_ITEMS], val[RS_ITEMS], rank[RS_ITEMS];
#pragma unroll
  for (int it = 0; it < RS_ITEMS; ++it) {
    int e = rs_elem(it);
    bool in = e < n;
    key[it] = in ? keys_in[e] : 0xFFFFFFFFu;
    val[it] = in ? vals_in[e] : 0u;
  }

  // global offset of digit d (= threadIdx.x) for this block:
  //   sum over all blocks of digits < d   +   sum over earlier blocks of digit d
  {
    const int d = threadIdx.x;
    uint32_t below = 0, total = 0;
    const int b = blockIdx.x;
    int bb = 0;
    for (; bb + 4 <= nblocks; bb += 4) {
      uint32_t t0 = hist[(size_t)(bb + 0) * RS_RADIX + d];
      uint32_t t1 = hist[(size_t)(bb + 1) * RS_RADIX + d];
      uint32_t t2 = hist[(size_t)(bb + 2) * RS_RADIX + d];
      uint32_t t3 = hist[(size_t)(bb + 3) * RS_RADIX + d];
      total += t0 + t1 + t2 + t3;
      below += (bb + 0 < b ? t0 : 0) + (bb + 1 < b ? t1 : 0) + (bb + 2 < b ? t2 : 0) + (bb + 3 < b ? t3 : 0);
    }
    for (; bb < nblocks; ++bb) {
      uint32_t t0 = hist[(size_t)bb * RS_RADIX + d];
      total += t0;
      below += (bb < b) ? t0 : 0;
    }
    uint32_t inc = wave_incl_scan_u32(total);
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    uint32_t woff = 0;
#pragma unroll
    for (int k = 0; k < RS_WAVES; ++k) woff += (k < w) ? wsum[k] : 0;
    gbase[d] = woff + inc - total + below;
  }
  __syncthreads();

  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
  for (int it = 0; it < RS_ITEMS; ++it) {
    int e = rs_elem(it);
    bool in = e < n;
    uint32_t d = (key[it] >> shift) & 0xFF;
    // lanes of this wave holding the same digit (out-of-range lanes excluded)
    unsigned long long same = __ballot(in);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      unsigned long long bal = __ballot((d >> b) & 1);
      same &= ((d >> b) & 1) ? bal : ~bal;
    }
    uint32_t before = (uint32_t)__popcll(same & lt_mask);
    volatile uint32_t* slot = &cnt[w][d];
    uint32_t old = *slot;                     // same value for every lane sharing d
    rank[it] = old + before;
    if (in && before == 0) *slot = old + (uint32_t)__popcll(same);   // leader updates
  }
  __syncthreads();
  // digit `threadIdx.x`: turn per-wave counts into per-wave bases (+ global block offset)
  {
    uint32_t g = gbase[threadIdx.x];
#pragma unroll
    for (int k = 0; k < RS_WAVES; ++k) {
      uint32_t t = cnt[k][threadIdx.x];
      cnt[k][threadIdx.x] = g;
      g += t;
    }
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < RS_ITEMS; ++it) {
    int e = rs_elem(it);
    if (e < n) {
      uint32_t d = (key[it] >> shift) & 0xFF;
      uint32_t pos = cnt[w][d] + rank[it];
      keys_out[pos] = key[it];
      vals_out[pos] = val[it];
    }
  }
}

// Sorts indices 0..n-1 by (key asc, index asc) = (score desc, index asc), B images in the same launches
// (blockIdx.y = image).  Per image: keys_a holds the keys on entry; keys_a/vals_a/keys_b/vals_b: n uint32 each; hist:
// odet_sort_hist_entries(n) uint32; skip: the image's "done" word (nullable).  After the 4 passes the result is in
// vals_a (and keys_a is sorted too).
int odet_sort_keys_desc_batch(int n, int B, const OdetSortImage* imgs, hipStream_t st) {
  if (B < 1 || B > ODET_MAX_BATCH) return odet_set_error(ODET_E_INVALID, "odet_sort_keys_desc_batch: batch %d out of range", B);
  RsBatch bt;
  for (int i = 0; i < ODET_MAX_BATCH; ++i) {
    const OdetSortImage& a = imgs[i < B ? i : 0];
    bt.v[i].ka = a.keys_a; bt.v[i].va = a.vals_a; bt.v[i].kb = a.keys_b; bt.v[i].vb = a.vals_b;
    bt.v[i].hist = a.hist; bt.v[i].skip = a.skip;
  }
  const int nblocks = (n + RS_TILE - 1) / RS_TILE;
  const dim3 grid(nblocks, B);
  hipLaunchKernelGGL(k_rs_init, grid, dim3(RS_BLOCK), 0, st, bt, n);
  ODET_LAUNCH_CHECK();
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = pass * 8, flip = pass & 1;
    if (pass > 0) {
      hipLaunchKernelGGL(k_rs_hist, grid, dim3(RS_BLOCK), 0, st, bt, n, shift, flip);
      ODET_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_rs_scatter, grid, dim3(RS_BLOCK), 0, st, bt, n, shift, nblocks, flip);
    ODET_LAUNCH_CHECK();
  }
  return ODET_OK;
}

// one image; *sorted_vals points at the buffer holding the result (vals_a after an even number of passes)
int odet_sort_keys_desc(int n, uint32_t* keys_a, uint32_t* vals_a, uint32_t* keys_b, uint32_t* vals_b,
                        uint32_t* hist, const int32_t* skip, uint32_t** sorted_vals, hipStream_t st) {
  const OdetSortImage one{keys_a, vals_a, keys_b, vals_b, hist, skip};
  const int rc = odet_sort_keys_desc_batch(n, 1, &one, st);
  *sorted_vals = vals_a;
  return rc;
}

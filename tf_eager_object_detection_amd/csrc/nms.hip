// Exact greedy NMS (tf.image.non_max_suppression / NonMaxSuppressionV3 semantics), the fused
// RegionProposal path (model/region_proposal.py:55-81) and the fused FPN proposal stage
// (model/fpn/base_fpn_model.py:220-224 + :303-324).
//
// TF's kernel is a serial loop: pop the best remaining candidate, test it against every box kept
// so far, keep it if no IoU > threshold, stop at max_output.  The result only depends on the
// candidates in (score desc, index asc) order until max_output are kept, so the GPU form is:
//
//   1. k_rp_prepare   : one pass over all n candidates: (anchors in registers ->) fg softmax ->
//                       decode -> clip -> box, 32-bit order key, 12-bit key histogram.
//   2. radix SELECT of the best ~1.5 K candidates instead of sorting all n:
//      k_sel_hist2    : every block finds the threshold bin of the 12-bit histogram (prefix scan in
//                       the prologue), then histograms the next 12 key bits of that bin's members;
//      k_sel_compact  : finds the 24-bit threshold prefix and appends every candidate at or below
//                       it (wave64 ballot + one atomic per wave) -- a set closed under the order,
//                       count <= 4096 (the boundary bin is dropped whole if it would not fit);
//      k_sel_rank     : orders those (key, index) pairs by counting (rank = number of smaller
//                       pairs; 64 candidates x 16 range slices per workgroup, list broadcast from
//                       LDS) and gathers their corner-normalised boxes.
//   3. k_nms_mask     : pairwise-suppression bit matrix of the chunk on the whole chip: one
//                       workgroup per 64x64 tile of the LOWER triangle, four waves x 16 columns,
//                       stored word-major  Lt[word b][candidate j]  (bit i: candidate 64b+i, earlier
//                       in score order, suppresses j if it is kept) so that the scan reads it
//                       coalesced; the diagonal tile is stored row-wise (who do I suppress).
//   4. k_nms_scan     : ONE workgroup, thread = candidate.  64-candidate blocks are resolved in
//                       order by the wave that owns them (ctz over the alive ballot + v_readlane of
//                       the diagonal rows, only candidates that actually suppress a live later one
//                       cost a serial step); after each block every later candidate folds
//                       Lt[b][j] & kept[b] into its "suppressed" flag (loads prefetched two blocks
//                       ahead).  One barrier per block, early exit at max_output.  The tail writes
//                       the kept boxes in score order and, for the FPN stage, assigns pyramid levels
//                       (stable partition) in the same launch.
//   5. Only if the chunk did not reach max_output: full radix sort (sort.hip) and further chunks
//      of 4096 candidates (k_nms_gather -> k_nms_cross vs. the boxes kept so far -> mask -> scan).
//
// Every decision is the predicate TF evaluates (d_iou_gt), so kept indices are identical to the
// serial algorithm, including the early stop at max_output.
#include <stddef.h>
#include <string.h>

#include <algorithm>
#include <mutex>

#include "odet_internal.h"

#define NMS_CHUNK 4096
#define NMS_SEL_MAX (2 * NMS_CHUNK)   // most candidates one radix selection hands over (first chunk + one more)
#define NMS_WORDS (NMS_CHUNK / 64)
#define SEL_BINS 4096
#define SEL_REPL 8   // copies of the first-level histogram (hot bins: same-address atomics serialise)
#define SEL_BLOCK 256
#define SEL_ITEMS 8
#define SEL_TILE (SEL_BLOCK * SEL_ITEMS)
#define PREP_ITEMS 2
#define PREP_TILE (256 * PREP_ITEMS)

typedef unsigned long long u64;

// Up to this many 64-candidate blocks the whole strictly-lower suppression matrix of a chunk is staged
// in LDS by the scan (276 tiles x 512 B = 138 KiB of the 160 KiB): its walk never waits for memory.
#define SCAN_LDS_BLOCKS 24
#define SCAN_LDS_CAND (SCAN_LDS_BLOCKS * 64)
#define SCAN_LDS_WORDS (SCAN_LDS_BLOCKS * (SCAN_LDS_BLOCKS - 1) / 2 * 64)
#define SCAN_LDS_SLACK_WORDS (8 * 64)   // unpredicated reads of the last wave may run past the image
#define SCAN_DYN_LDS ((SCAN_LDS_WORDS + SCAN_LDS_SLACK_WORDS) * 8)
// first word of column block cb inside the packed image (rows (cb+1)*64 .. nblk*64-1 follow each other)
#define SCAN_MAT_OFF(cb, nblk) (64 * ((cb) * ((nblk)-1) - (cb) * ((cb)-1) / 2))

struct NmsState {
  int32_t n_invalid;     // scores TF would not push into its heap (NaN, <= lowest float)
  int32_t kept;          // boxes kept so far
  int32_t pos;           // candidates (in score order) consumed so far
  int32_t done;          // kept == K or pos == n_valid
  int32_t chunk_m;       // size of the current chunk
  int32_t sel_count;     // candidates appended by k_sel_compact
  int32_t reserved0;
  int32_t sel_b1;        // threshold bin of the first 12 key bits
  uint32_t sel_below1;   // candidates in bins < sel_b1
  // tie split (k_sel_tie_hist / k_sel_tie_take): the boundary bin (24-bit prefix sel_b1:sel_b2) did not fit the
  // selection and was left out by k_sel_compact; sel_below = candidates strictly below it
  int32_t sel_b2;
  int32_t sel_overflow;
  uint32_t sel_below;
  int32_t pad[4];
};

// Starts the workspace.  Zeroed by k_zero_headers at the start of a call -- or, for callers that promise a workspace
// that was zero-filled once and is only ever used by this library (odet_fpn_step_t.ws_rpn_clean), left clean by the
// call itself: hist0 / hist1 are zeroed again by k_sel_compact (k_sel_hist2's blocks were their last readers), hist2 by
// k_sel_rank (after k_sel_compact), n_invalid_acc is moved into the state by k_sel_hist2, which also resets the state.
#define SEL_COARSE 64                   // coarse bins: 64 fine bins each
struct NmsHeader {
  NmsState st;
  int32_t n_invalid_acc;               // fed by k_rp_prepare; k_sel_hist2 moves it into st.n_invalid
  int32_t pad_acc[15];
  // sums of 64 consecutive first-level bins: with them a block finds the threshold bin from 2 x 64 x SEL_REPL counters
  // instead of 4096 x SEL_REPL (round 4: k_sel_find1, a launch of its own for one workgroup's work, is gone)
  uint32_t hist0[SEL_REPL][SEL_COARSE];
  uint32_t hist1[SEL_REPL][SEL_BINS];  // replica r is fed by blocks with blockIdx % SEL_REPL == r
  uint32_t hist2[SEL_BINS];
  uint32_t hist3[256];                 // tie split: the last 8 key bits inside the boundary bin (zeroed again by k_sel_rank)
};

// ------------------------------------------------------------------------- 1. prepare -------
enum { PREP_NMS = 0, PREP_RP = 1, PREP_FPN = 2, PREP_FRCNN = 3 };

struct PrepParams {          // pointer tables: one entry per image of the batch (blockIdx.y)
  int n;
  PerImg<const float4*> boxes_in;   // PREP_NMS: boxes (not rewritten); PREP_RP: anchors
  PerImg<const float*> deltas;      // PREP_RP / PREP_FPN: [n,4]
  PerImg<const float*> scores;      // PREP_NMS / PREP_RP
  PerImg<const float2*> logits;     // PREP_FPN: (bg, fg) pairs; PREP_FRCNN: rows [A bg | A fg] (read as floats)
  Vec4 means, stds;
  float wmax, hmax;
  PerImg<float4*> boxes_out;        // PREP_RP / PREP_FPN: decoded + clipped boxes
  PerImg<uint32_t*> keys;
  PerImg<NmsHeader*> hdr;
  FpnAnchorParams fpn;      // PREP_FPN; PREP_FRCNN: A, fw[0], stride[0], wh[0..4A) = base anchors
};

// zero the per-call header of every image of the batch (state + histograms)
__global__ void __launch_bounds__(256) k_zero_headers(PerImg<NmsHeader*> hdr_) {
  uint4* d = reinterpret_cast<uint4*>(hdr_.v[blockIdx.y]);
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < (int)(sizeof(NmsHeader) / 16)) d[i] = make_uint4(0, 0, 0, 0);
}

// Box of candidate e of image img -- decoded LAZILY: the order key needs only the score, and of the 267 069
// anchors of a pyramid only the few thousand candidates the selection hands to NMS ever need their box
// (anchor in registers, 16 B of deltas, decode + clip: the same device functions, the same bits as a full
// decode pass, which cost 32 of the 44 bytes per anchor of this stage and a float64 exp pair per anchor).
template <int MODE>
__device__ __forceinline__ float4 d_prep_box(const PrepParams& p, int img, int e) {
  if (MODE == PREP_NMS) return p.boxes_in.v[img][e];
  float4 a;
  if (MODE == PREP_FPN) {
    a = d_fpn_anchor(p.fpn, e);                                    // base_fpn_model.py:220 / :163-186
  } else if (MODE == PREP_FRCNN) {
    // anchor_generator.py:46-60 generate_by_anchor_base_tf, in registers
    const int ai = e % p.fpn.A, cell = e / p.fpn.A;
    const int x = cell % p.fpn.fw[0], y = cell / p.fpn.fw[0];
    const float sx = (float)(x * p.fpn.stride[0]), sy = (float)(y * p.fpn.stride[0]);
    a = make_float4(p.fpn.wh[ai * 4 + 0] + sx, p.fpn.wh[ai * 4 + 1] + sy, p.fpn.wh[ai * 4 + 2] + sx,
                    p.fpn.wh[ai * 4 + 3] + sy);
  } else {
    a = p.boxes_in.v[img][e];                                      // PREP_RP: anchors given
  }
  const float4 d = reinterpret_cast<const float4*>(p.deltas.v[img])[e];
  const float d0 = d.x * p.stds.v[0] + p.means.v[0];               // bbox_transform.py:37
  const float d1 = d.y * p.stds.v[1] + p.means.v[1];
  const float d2 = d.z * p.stds.v[2] + p.means.v[2];
  const float d3 = d.w * p.stds.v[3] + p.means.v[3];
  float4 b = d_decode_box(a, d0, d1, d2, d3);                      // region_proposal.py:59
  return d_clip_box(b, 0.0f, p.wmax, p.hmax);                      // :63
}

__device__ __forceinline__ float4 d_candidate_box(const PrepParams& p, int mode, int img, int e) {
  switch (mode) {
    case PREP_NMS: return d_prep_box<PREP_NMS>(p, img, e);
    case PREP_RP: return d_prep_box<PREP_RP>(p, img, e);
    case PREP_FRCNN: return d_prep_box<PREP_FRCNN>(p, img, e);
    default: return d_prep_box<PREP_FPN>(p, img, e);
  }
}

template <int MODE>
__global__ void __launch_bounds__(256) k_rp_prepare(PrepParams p) {
  const int img = blockIdx.y;
  const float* __restrict__ in_scores = p.scores.v[img];
  const float2* __restrict__ in_logits = p.logits.v[img];
  uint32_t* __restrict__ out_keys = p.keys.v[img];
  NmsHeader* hdr = p.hdr.v[img];
  __shared__ uint32_t h[SEL_BINS];
  __shared__ uint32_t h0[SEL_COARSE];          // sums of 64 consecutive bins, fed beside them (k_sel_hist2's prologue)
  for (int k = threadIdx.x; k < SEL_BINS; k += 256) h[k] = 0;
  if (threadIdx.x < SEL_COARSE) h0[threadIdx.x] = 0;
  // all loads of the tile first (independent, in flight together), then the arithmetic
  float sc[PREP_ITEMS];
  float2 lg[PREP_ITEMS];
#pragma unroll
  for (int it = 0; it < PREP_ITEMS; ++it) {
    const int e = blockIdx.x * PREP_TILE + it * 256 + threadIdx.x;
    const bool in = e < p.n;
    const int ee = in ? e : 0;
    if (MODE == PREP_FPN) {
      lg[it] = in_logits[ee];
    } else if (MODE == PREP_FRCNN) {
      // base_faster_rcnn_model.py:149-152: per location [A bg | A fg]
      const int loc = ee / p.fpn.A, a = ee - loc * p.fpn.A;
      const float* row = reinterpret_cast<const float*>(in_logits) + (size_t)loc * 2 * p.fpn.A;
      lg[it] = make_float2(row[a], row[p.fpn.A + a]);
    } else {
      sc[it] = in_scores[ee];
    }
  }
  __syncthreads();
  int invalid = 0;
#pragma unroll
  for (int it = 0; it < PREP_ITEMS; ++it) {
    const int e = blockIdx.x * PREP_TILE + it * 256 + threadIdx.x;
    if (e < p.n) {
      float s;
      if (MODE == PREP_FPN || MODE == PREP_FRCNN) s = d_fg_prob(lg[it].x, lg[it].y);   // base_fpn_model.py:223
      else s = sc[it];
      const bool valid = s > -3.402823466e+38f;   // NonMaxSuppressionV3: score > score_threshold (= lowest)
      const uint32_t k = valid ? ~d_float_asc_key(s) : 0xFFFFFFFFu;
      out_keys[e] = k;
      invalid += valid ? 0 : 1;
      atomicAdd(&h[k >> 20], 1u);
      atomicAdd(&h0[k >> 26], 1u);
    }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < SEL_BINS; k += 256) {
    const uint32_t c = h[k];
    if (c) atomicAdd(&hdr->hist1[blockIdx.x % SEL_REPL][k], c);
  }
  if (threadIdx.x < SEL_COARSE) {
    const uint32_t c = h0[threadIdx.x];
    if (c) atomicAdd(&hdr->hist0[blockIdx.x % SEL_REPL][threadIdx.x], c);
  }
  if (invalid) atomicAdd(&hdr->n_invalid_acc, invalid);            // rare (k_sel_hist2 moves it into the state)
}

// ------------------------------------------------------------------------- 2. select --------
// Smallest bin whose inclusive prefix count reaches `target` (1 <= target), and the count below
// it.  256 threads x 16 bins.  result[0] = bin, result[1] = below, result[2] = count of the bin.
// target > total -> treated as total.
template <int REPL>
__device__ __forceinline__ void sel_find(const uint32_t* __restrict__ hist, uint32_t target, uint32_t* result,
                                         int* lds17) {
  uint32_t v[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) v[q] = 0;
#pragma unroll
  for (int r = 0; r < REPL; ++r) {
    const uint4* h4 = reinterpret_cast<const uint4*>(hist + (size_t)r * SEL_BINS) + threadIdx.x * 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint4 t = h4[q];
      v[q * 4 + 0] += t.x; v[q * 4 + 1] += t.y; v[q * 4 + 2] += t.z; v[q * 4 + 3] += t.w;
    }
  }
  uint32_t s = 0;
#pragma unroll
  for (int q = 0; q < 16; ++q) s += v[q];
  int total;
  const uint32_t excl = (uint32_t)block_excl_scan((int)s, lds17, &total);
  if ((uint32_t)total < target) target = (uint32_t)total;
  if (threadIdx.x == 0 && total == 0) { result[0] = SEL_BINS - 1; result[1] = 0; result[2] = 0; }
  if (s > 0 && excl < target && target <= excl + s) {
    uint32_t run = excl;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      if (run < target && target <= run + v[q]) {
        result[0] = threadIdx.x * 16 + q; result[1] = run; result[2] = v[q];
      }
      run += v[q];
    }
  }
  __syncthreads();
}

// First-level threshold bin (the bin holding the target-th best key, and the count below it), found by EVERY block of
// k_sel_hist2 in its prologue from the coarse sums: wave 0 scans the 64 coarse bins, then the 64 fine bins of the coarse
// bin that holds the target -- 2 x 64 x SEL_REPL counters from L2 (a block re-deriving it from all 4096 x SEL_REPL fine
// counters cost more than a launch of its own, which is what k_sel_find1 was until round 4: 5.5 us + a kernel boundary
// per image in the latency arrangement for one workgroup's work).  res[0] = bin, res[1] = count below.
__device__ __forceinline__ void sel_find1_coarse(const NmsHeader* hdr, uint32_t target, uint32_t* res) {
  if (threadIdx.x < 64) {
    const int t = threadIdx.x;
    int c = 0;
#pragma unroll
    for (int r = 0; r < SEL_REPL; ++r) c += (int)hdr->hist0[r][t];
    const int inc = wave_incl_scan(c);
    const int total = __shfl(inc, 63);
    if (total == 0) {
      if (t == 0) { res[0] = SEL_BINS - 1; res[1] = 0; }
    } else {
      const int tgt = (int)(target > (uint32_t)total ? (uint32_t)total : target);
      const u64 hitc = __ballot(c > 0 && inc - c < tgt && tgt <= inc);
      const int cb = __builtin_ctzll(hitc);
      const int below_c = __shfl(inc - c, cb);
      int v = 0;
#pragma unroll
      for (int r = 0; r < SEL_REPL; ++r) v += (int)hdr->hist1[r][cb * 64 + t];
      const int inc2 = wave_incl_scan(v);
      const u64 hitf = __ballot(v > 0 && below_c + inc2 - v < tgt && tgt <= below_c + inc2);
      const int fb = __builtin_ctzll(hitf);
      const int below_f = __shfl(inc2 - v, fb);
      if (t == 0) { res[0] = (uint32_t)(cb * 64 + fb); res[1] = (uint32_t)(below_c + below_f); }
    }
  }
  __syncthreads();
}

__global__ void __launch_bounds__(SEL_BLOCK) k_sel_hist2(PerImg<NmsHeader*> hdr_, PerImg<const uint32_t*> keys_, int n,
                                                         uint32_t target) {
  NmsHeader* hdr = hdr_.v[blockIdx.y];
  const uint32_t* __restrict__ keys = keys_.v[blockIdx.y];
  __shared__ uint32_t h[SEL_BINS];
  __shared__ uint32_t res[2];
  // the tile's keys first (in flight while wave 0 finds the threshold bin)
  uint32_t key[SEL_ITEMS];
#pragma unroll
  for (int it = 0; it < SEL_ITEMS; ++it) {
    const int e = blockIdx.x * SEL_TILE + it * SEL_BLOCK + threadIdx.x;
    key[it] = (e < n) ? keys[e] : 0u;
  }
  sel_find1_coarse(hdr, target, res);
  const uint32_t b1 = res[0];
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    // the state of a new job (a workspace kept clean between calls is not zeroed by a launch of its own); the other
    // blocks of this launch use their own copy of b1, the later kernels read it from here
    NmsState z;
    memset(&z, 0, sizeof(z));
    z.n_invalid = hdr->n_invalid_acc;
    z.sel_b1 = (int32_t)b1; z.sel_below1 = res[1];
    hdr->st = z;
    hdr->n_invalid_acc = 0;
  }
  bool match = false;
#pragma unroll
  for (int it = 0; it < SEL_ITEMS; ++it) {
    const int e = blockIdx.x * SEL_TILE + it * SEL_BLOCK + threadIdx.x;
    match = match || (e < n && (key[it] >> 20) == b1);
  }
  if (!__syncthreads_or(match ? 1 : 0)) return;
  for (int k = threadIdx.x; k < SEL_BINS; k += SEL_BLOCK) h[k] = 0;
  __syncthreads();
#pragma unroll
  for (int it = 0; it < SEL_ITEMS; ++it) {
    const int e = blockIdx.x * SEL_TILE + it * SEL_BLOCK + threadIdx.x;
    if (e < n && (key[it] >> 20) == b1) atomicAdd(&h[(key[it] >> 8) & 0xFFFu], 1u);
  }
  __syncthreads();
  for (int k = threadIdx.x; k < SEL_BINS; k += SEL_BLOCK) {
    const uint32_t c = h[k];
    if (c) atomicAdd(&hdr->hist2[k], c);
  }
}

__global__ void __launch_bounds__(SEL_BLOCK) k_sel_compact(PerImg<NmsHeader*> hdr_, PerImg<const uint32_t*> keys_, int n,
                                                           uint32_t target, uint32_t limit, PerImg<u64*> cand_) {
  NmsHeader* hdr = hdr_.v[blockIdx.y];
  const uint32_t* __restrict__ keys = keys_.v[blockIdx.y];
  u64* __restrict__ cand = cand_.v[blockIdx.y];
  __shared__ uint32_t res[3];
  __shared__ int lds17[17];
  const uint32_t b1 = (uint32_t)hdr->st.sel_b1;
  const uint32_t below1 = hdr->st.sel_below1;
  // k_sel_hist2's blocks were the last readers of the first-level histograms: leave them clean for the next call
  {
    static_assert(offsetof(NmsHeader, hist1) == offsetof(NmsHeader, hist0) + sizeof(uint32_t) * SEL_REPL * SEL_COARSE, "layout");
    uint4* z = reinterpret_cast<uint4*>(&hdr->hist0[0][0]);
    for (int i = blockIdx.x * SEL_BLOCK + threadIdx.x; i < SEL_REPL * (SEL_COARSE + SEL_BINS) / 4; i += gridDim.x * SEL_BLOCK)
      z[i] = make_uint4(0, 0, 0, 0);
  }
  uint32_t total_n = (uint32_t)n;
  uint32_t t2 = (target > total_n ? total_n : target) - below1;   // >= 1 by construction of b1
  sel_find<1>(hdr->hist2, t2, res, lds17);
  const uint32_t b2 = res[0];
  // The selected set must be closed under the order, so the boundary bin (24-bit prefix b1:b2) is
  // taken whole or not at all: whole if the set still fits `limit` candidates (the capacity of the
  // chunk-0 scan), otherwise the chunk is the (smaller, still valid) set strictly below it and the
  // fallback path picks up from there.
  const bool incl = (below1 + res[1] + res[2]) <= limit;
  if (blockIdx.x == 0 && threadIdx.x == 0) {               // for the tie split (wide selections only)
    hdr->st.sel_b2 = (int32_t)b2;
    hdr->st.sel_overflow = incl ? 0 : 1;
    hdr->st.sel_below = below1 + res[1];
  }
  // block-local append (LDS counter, one LDS atomic per wave and item), then ONE global atomic per block
  __shared__ u64 stage[SEL_TILE];
  __shared__ int s_cnt, s_base;
  if (threadIdx.x == 0) s_cnt = 0;
  uint32_t key[SEL_ITEMS];
#pragma unroll
  for (int it = 0; it < SEL_ITEMS; ++it) {
    const int e = blockIdx.x * SEL_TILE + it * SEL_BLOCK + threadIdx.x;
    key[it] = (e < n) ? keys[e] : 0xFFFFFFFFu;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const u64 lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
  for (int it = 0; it < SEL_ITEMS; ++it) {
    const int e = blockIdx.x * SEL_TILE + it * SEL_BLOCK + threadIdx.x;
    const uint32_t k = key[it];
    const uint32_t p1 = k >> 20, p2 = (k >> 8) & 0xFFFu;
    const bool sel = (e < n) && (p1 < b1 || (p1 == b1 && (p2 < b2 || (incl && p2 == b2))));
    const u64 bal = __ballot(sel);
    if (bal) {
      int base = 0;
      if (lane == __builtin_ctzll(bal)) base = atomicAdd(&s_cnt, (int)__popcll(bal));
      base = __shfl(base, __builtin_ctzll(bal));
      if (sel) stage[base + (int)__popcll(bal & lt_mask)] = ((u64)k << 32) | (uint32_t)e;
    }
  }
  __syncthreads();
  const int c = s_cnt;
  if (c == 0) return;
  if (threadIdx.x == 0) s_base = atomicAdd(&hdr->st.sel_count, c);
  __syncthreads();
  const int gb = s_base;
  for (int i = threadIdx.x; i < c; i += SEL_BLOCK)
    if (gb + i < NMS_SEL_MAX) cand[gb + i] = stage[i];
}

// ---- tie split: a boundary bin that does not fit ------------------------------------------------------------------------
// A saturated RPN (softmax == 1.0f for tens of thousands of anchors: the fg - bg margin is beyond float32's reach -- the
// random-init float16 detector of the benchmark, and any confident trained one) puts thousands of EQUAL keys at the very
// top of the order: the boundary bin of the selection holds more candidates than the selection may take, k_sel_compact
// leaves it out whole, and the chunks from the selection come up empty.  For the wide selection of sync-free jobs with
// more than one chunk the bin is split exactly, in (key, index) order -- the declared tie rule, SURVEY H2:
//   k_sel_tie_hist : histogram of the LAST 8 key bits inside the boundary bin, per block (index ranges in order) and summed
//   k_sel_tie_take : boundary value b3 = where the running count reaches the quota; everything below b3 is taken, of the
//                    keys EQUAL to b3 the first q3 in index order: a block knows how many ties the blocks before it hold
//                    (per-block counts) and ranks its own by ballots.
// Both exit at once when the bin fitted (st.sel_overflow == 0).  The selection stays an exact prefix of the order.
__global__ void __launch_bounds__(SEL_BLOCK) k_sel_tie_hist(PerImg<NmsHeader*> hdr_, PerImg<const uint32_t*> keys_, int n,
                                                            PerImg<unsigned short*> bh_) {
  NmsHeader* hdr = hdr_.v[blockIdx.y];
  if (!hdr->st.sel_overflow) return;
  const uint32_t* __restrict__ keys = keys_.v[blockIdx.y];
  unsigned short* __restrict__ bh = bh_.v[blockIdx.y] + (size_t)blockIdx.x * 256;
  __shared__ uint32_t h[256];
  const uint32_t pref = ((uint32_t)hdr->st.sel_b1 << 12) | (uint32_t)hdr->st.sel_b2;
  h[threadIdx.x] = 0;                                       // (SEL_BLOCK == 256)
  __syncthreads();
#pragma unroll
  for (int it = 0; it < SEL_ITEMS; ++it) {
    const int e = blockIdx.x * SEL_TILE + it * SEL_BLOCK + threadIdx.x;
    if (e < n) {
      const uint32_t k = keys[e];
      if ((k >> 8) == pref) atomicAdd(&h[k & 0xFFu], 1u);
    }
  }
  __syncthreads();
  const uint32_t c = h[threadIdx.x];
  bh[threadIdx.x] = (unsigned short)c;
  if (c) atomicAdd(&hdr->hist3[threadIdx.x], c);
}

__global__ void __launch_bounds__(SEL_BLOCK) k_sel_tie_take(PerImg<NmsHeader*> hdr_, PerImg<const uint32_t*> keys_, int n,
                                                            uint32_t limit, PerImg<const unsigned short*> bh_,
                                                            PerImg<u64*> cand_) {
  NmsHeader* hdr = hdr_.v[blockIdx.y];
  if (!hdr->st.sel_overflow) return;
  const uint32_t below = hdr->st.sel_below;
  if (below >= limit) return;
  const uint32_t quota = limit - below;                    // candidates of the boundary bin the selection still takes
  const uint32_t* __restrict__ keys = keys_.v[blockIdx.y];
  const unsigned short* __restrict__ bh = bh_.v[blockIdx.y];
  u64* __restrict__ cand = cand_.v[blockIdx.y];
  __shared__ int lds17[17];
  __shared__ uint32_t s_b3, s_q3, s_pre;
  __shared__ int s_wave[4];
  __shared__ u64 stage[SEL_TILE];
  __shared__ int s_cnt, s_base;
  const uint32_t pref = ((uint32_t)hdr->st.sel_b1 << 12) | (uint32_t)hdr->st.sel_b2;
  // boundary value: running count over the 256 last-byte values (thread = value)
  const uint32_t hv = hdr->hist3[threadIdx.x];
  int total;
  const uint32_t excl = (uint32_t)block_excl_scan((int)hv, lds17, &total);
  if (threadIdx.x == 0) { s_b3 = 256; s_q3 = 0; s_pre = 0; s_cnt = 0; }
  __syncthreads();
  if (hv > 0 && excl < quota && quota <= excl + hv) { s_b3 = threadIdx.x; s_q3 = quota - excl; }
  __syncthreads();
  const uint32_t b3 = s_b3, q3 = s_q3;                     // (b3 == 256: the whole bin fits after all -- not reached)
  // ties of value b3 in the blocks before this one
  if (b3 < 256) {
    uint32_t mine = 0;
    for (int b = threadIdx.x; b < (int)blockIdx.x; b += SEL_BLOCK) mine += bh[(size_t)b * 256 + b3];
    int tot2;
    block_excl_scan((int)mine, lds17, &tot2);
    if (threadIdx.x == 0) s_pre = (uint32_t)tot2;
  }
  __syncthreads();
  uint32_t run = s_pre;                                    // ties of value b3 before the current item (block-uniform)
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const u64 lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  for (int it = 0; it < SEL_ITEMS; ++it) {
    const int e = blockIdx.x * SEL_TILE + it * SEL_BLOCK + threadIdx.x;
    const uint32_t k = (e < n) ? keys[e] : 0xFFFFFFFFu;
    const bool inbin = (e < n) && (k >> 8) == pref;
    const uint32_t v = k & 0xFFu;
    const bool tie = inbin && v == b3;
    const u64 tb = __ballot(tie);
    if (lane == 0) s_wave[wv] = (int)__popcll(tb);
    __syncthreads();
    uint32_t before = run;
    for (int q = 0; q < wv; ++q) before += (uint32_t)s_wave[q];
    const uint32_t my_rank = before + (uint32_t)__popcll(tb & lt_mask);
    const uint32_t all4 = (uint32_t)(s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3]);
    const bool sel = inbin && (v < b3 || (tie && my_rank < q3));
    const u64 bal = __ballot(sel);
    if (bal) {
      int base = 0;
      if (lane == __builtin_ctzll(bal)) base = atomicAdd(&s_cnt, (int)__popcll(bal));
      base = __shfl(base, __builtin_ctzll(bal));
      if (sel) stage[base + (int)__popcll(bal & lt_mask)] = ((u64)k << 32) | (uint32_t)e;
    }
    run += all4;
    __syncthreads();
  }
  const int c = s_cnt;
  if (c == 0) return;
  if (threadIdx.x == 0) s_base = atomicAdd(&hdr->st.sel_count, c);
  __syncthreads();
  const int gb = s_base;
  for (int i = threadIdx.x; i < c; i += SEL_BLOCK)
    if (gb + i < NMS_SEL_MAX) cand[gb + i] = stage[i];
}

// chunk 0: order the selected candidates by counting (all (key, index) pairs are distinct, so the
// rank of a pair = number of smaller pairs is a permutation) and gather their boxes.  Workgroup =
// 64 candidates (lane) x 16 slices of the comparison range (wave); the pair list is staged in LDS and
// every wave reads it as a broadcast.
#define RANK_THREADS 1024
__global__ void __launch_bounds__(RANK_THREADS) k_sel_rank(PerImg<NmsHeader*> hdr_, int n, int cap0, PerImg<const u64*> cand_,
                                                          PrepParams prep, int mode, PerImg<uint32_t*> sel_idx_,
                                                          PerImg<float4*> sboxes_, PerImg<float4*> sorig_) {
  NmsHeader* hdr = hdr_.v[blockIdx.y];
  const u64* __restrict__ cand = cand_.v[blockIdx.y];
  uint32_t* __restrict__ sel_idx = sel_idx_.v[blockIdx.y];
  float4* __restrict__ sboxes = sboxes_.v[blockIdx.y];
  float4* __restrict__ sorig = sorig_.v[blockIdx.y];
  __shared__ u64 all[NMS_CHUNK];
  __shared__ int part[16][64];
  NmsState* st = &hdr->st;
  // the selection may hold more than the first chunk takes (cap0): the rest, ranked here as well, is the next
  // sync-free chunk's (k_nms_gather with the selection as its order)
  const int cnt = min(st->sel_count, NMS_SEL_MAX);
  if (blockIdx.x == 0 && threadIdx.x == 0) st->chunk_m = min(min(cnt, cap0), n - st->n_invalid);
  if (blockIdx.x == 0) {     // k_sel_compact was the last reader of the second-level histogram: leave it clean
    uint4* z = reinterpret_cast<uint4*>(hdr->hist2);
    for (int i = threadIdx.x; i < SEL_BINS / 4; i += RANK_THREADS) z[i] = make_uint4(0, 0, 0, 0);
    if (threadIdx.x < 256) hdr->hist3[threadIdx.x] = 0;      // (and the tie split's, whose last reader ran before this launch)
  }
  if (blockIdx.x * 64 >= cnt) return;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + lane;
  const u64 mine = (i < cnt) ? cand[i] : ~0ull;
  // the candidate's box does not depend on its rank: wave 0 requests its deltas NOW (decode + clip: a dependent global round trip
  // and a float64 exp pair) so that they travel while all 16 waves count, instead of after the count (round 6: the latency
  // arrangement's chain is one round trip shorter)
  float4 bx = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  if (w == 0 && i < cnt) bx = d_candidate_box(prep, mode, blockIdx.y, (int)(uint32_t)mine);
  int c = 0;
  for (int t0 = 0; t0 < cnt; t0 += NMS_CHUNK) {          // the pair list goes through LDS a tile at a time
    const int tc = min(NMS_CHUNK, cnt - t0);
    __syncthreads();
    for (int k = threadIdx.x; k < tc; k += RANK_THREADS) all[k] = cand[t0 + k];
    __syncthreads();
    const int per = (tc + 15) >> 4;
    const int lo = w * per, hi = min(tc, lo + per);
    for (int j = lo; j < hi; ++j) c += (all[j] < mine) ? 1 : 0;
  }
  part[w][lane] = c;
  __syncthreads();
  if (w == 0 && i < cnt) {
    int r = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) r += part[k][lane];
    const uint32_t idx = (uint32_t)mine;
    sel_idx[r] = idx;
    // invalid scores (key 0xFFFFFFFF) sort last; rows >= chunk_m are never read
    if (r < cap0) {
      sboxes[r] = d_norm_box(bx);
      sorig[r] = bx;
    }
  }
}

// later chunks: gather the chunk's boxes in sorted order (corner-normalised) -----------------
// from_sel: the order is the ranked radix selection (sel_idx), which ends at sel_count; otherwise the full sort.
__global__ void __launch_bounds__(256) k_nms_gather(PerImg<NmsState*> st_, int n, int cap, int from_sel,
                                                    PrepParams prep, int mode, PerImg<const uint32_t*> sorted_idx_,
                                                    PerImg<float4*> sboxes_, PerImg<float4*> sorig_) {
  NmsState* st = st_.v[blockIdx.y];
  const uint32_t* __restrict__ sorted_idx = sorted_idx_.v[blockIdx.y];
  float4* __restrict__ sboxes = sboxes_.v[blockIdx.y];
  float4* __restrict__ sorig = sorig_.v[blockIdx.y];
  const int pos = st->pos, nv = n - st->n_invalid;
  const int lim = from_sel ? min(nv, min(st->sel_count, NMS_SEL_MAX)) : nv;
  const int m = st->done ? 0 : max(0, min(cap, lim - pos));
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i == 0) st->chunk_m = m;
  if (i < m) {
    const float4 bx = d_candidate_box(prep, mode, blockIdx.y, (int)sorted_idx[pos + i]);
    sboxes[i] = d_norm_box(bx);
    sorig[i] = bx;
  }
}

// chunk candidates vs boxes kept by earlier chunks ---------------------------------------------
__global__ void __launch_bounds__(256) k_nms_cross(PerImg<const NmsState*> st_, PerImg<const float4*> sboxes_,
                                                   PerImg<const float4*> kept_boxes_, float thr,
                                                   PerImg<u64*> removed_init_) {
  const NmsState* st = st_.v[blockIdx.y];
  const float4* __restrict__ sboxes = sboxes_.v[blockIdx.y];
  const float4* __restrict__ kept_boxes = kept_boxes_.v[blockIdx.y];
  u64* __restrict__ removed_init = removed_init_.v[blockIdx.y];
  __shared__ float4 kb[256];
  __shared__ float ka[256];
  const int m = st->chunk_m, nk = st->kept;
  if (blockIdx.x * 256 >= m) return;          // nothing to do for this block (also: finished NMS)
  const int j = blockIdx.x * 256 + threadIdx.x;
  const float4 b = (j < m) ? sboxes[j] : make_float4(0, 0, 0, 0);
  const float area = d_box_area(b);
  bool sup = false;
  for (int k0 = 0; k0 < nk; k0 += 256) {
    __syncthreads();
    if (k0 + threadIdx.x < nk) {
      float4 t = kept_boxes[k0 + threadIdx.x];
      kb[threadIdx.x] = t;
      ka[threadIdx.x] = d_box_area(t);
    }
    __syncthreads();
    const int lim = min(256, nk - k0);
    for (int k = 0; k < lim; ++k) sup = sup || d_iou_gt(b, area, kb[k], ka[k], thr);
  }
  u64 bal = __ballot(sup && j < m);
  if ((threadIdx.x & 63) == 0) removed_init[j >> 6] = bal;
}

// ------------------------------------------------------------------------- 3. mask ----------
// 1-D grid over the lower-triangular tiles (rb, cb <= rb), row block major.  Workgroup = 4 waves,
// lane = row of the tile, wave w tests the row against columns [16w, 16w+16) of the column block.
__global__ void __launch_bounds__(256) k_nms_mask(PerImg<const NmsState*> st_, PerImg<const float4*> sboxes_, float thr,
                                                  PerImg<u64*> Lt_, PerImg<u64*> diag_, int packed) {
  const NmsState* st = st_.v[blockIdx.y];
  const float4* __restrict__ sboxes = sboxes_.v[blockIdx.y];
  u64* __restrict__ Lt = Lt_.v[blockIdx.y];
  u64* __restrict__ diag_up = diag_.v[blockIdx.y];
  const int t = blockIdx.x;
  int rb = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
  while ((rb + 1) * (rb + 2) / 2 <= t) ++rb;
  while (rb * (rb + 1) / 2 > t) --rb;
  const int cb = t - rb * (rb + 1) / 2;
  const int m = st->chunk_m;
  if (rb * 64 >= m) return;
  __shared__ float4 cbox[64];
  __shared__ float carea[64];
  __shared__ uint32_t part[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (threadIdx.x < 64) {
    const int col = cb * 64 + lane;
    const float4 c = (col < m) ? sboxes[col] : make_float4(0, 0, 0, 0);   // zero area never suppresses
    cbox[lane] = c;
    carea[lane] = d_box_area(c);
  }
  __syncthreads();
  const int row = rb * 64 + lane;
  const float4 r = (row < m) ? sboxes[row] : make_float4(0, 0, 0, 0);
  const float ra = d_box_area(r);
  uint32_t bits = 0;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int jj = w * 16 + j;
    const bool s = d_iou_gt(r, ra, cbox[jj], carea[jj], thr);
    bits |= s ? (1u << j) : 0u;
  }
  part[w][lane] = bits;
  __syncthreads();
  if (w == 0 && row < m) {
    const u64 word = (u64)part[0][lane] | ((u64)part[1][lane] << 16) | ((u64)part[2][lane] << 32) |
                     ((u64)part[3][lane] << 48);
    if (cb == rb) {
      // diagonal tile, row-wise: later candidates (bit j > lane) this row suppresses when kept
      const u64 later = (lane == 63) ? 0ull : (~0ull << (lane + 1));
      diag_up[row] = word & later;
    } else {
      // earlier candidates of block cb that suppress `row`; packed = the LDS image of k_nms_scan<true>
      if (packed) Lt[SCAN_MAT_OFF(cb, (m + 63) >> 6) + row - (cb + 1) * 64] = word;
      else Lt[(size_t)cb * NMS_CHUNK + row] = word;
    }
  }
}

// ------------------------------------------------------------------------- 4. scan ----------
struct AssignOut {      // optional fused _assign_levels (base_fpn_model.py:303-324) of the final RoIs
  float4* rois;         // null = disabled
  int32_t* level;
  int64_t* perm;
  int32_t* counts;
  int32_t* order;       // nullable: spatial processing order (see ScanParams::as_order)
  int min_level, max_level;
};

#define SCAN_THREADS 512
#define SCAN_WAVES (SCAN_THREADS / 64)
#define SCAN_Q 8      // candidates per thread: wave w owns candidates [512w, 512w+512) = blocks 8w..8w+7
#define SCAN_RING 4   // <false> path: column words of this many blocks are in flight per wave (registers)
#define SCAN_STAGE_ITEMS ((SCAN_LDS_WORDS / 2 + SCAN_THREADS - 1) / SCAN_THREADS)   // 16-B units per thread

__device__ __forceinline__ u64 rfl64(u64 v) {
  uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
  uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return ((u64)hi << 32) | lo;
}

// Resolve one 64-candidate block inside a wave.  alive: candidates not suppressed by earlier blocks
// (wave-uniform mask); d: diagonal row of the lane's candidate (later candidates of the block it
// suppresses when kept).  Only candidates that suppress a still-alive later one need a serial step; the
// rest of the alive ones are kept.  Returns the kept bits truncated to `room` survivors.
__device__ __forceinline__ u64 scan_resolve_block(u64 alive, u64 d, int room, int lane, u64 lt_lane, int* npop_out) {
  const uint32_t dlo = (uint32_t)d, dhi = (uint32_t)(d >> 32);
  for (;;) {
    const bool is_s = ((alive >> lane) & 1ull) && ((d & alive) != 0ull);
    const u64 supp = __ballot(is_s);
    if (supp == 0) break;
    const int i = __builtin_ctzll(supp);
    const uint32_t rlo = __builtin_amdgcn_readlane(dlo, i);
    const uint32_t rhi = __builtin_amdgcn_readlane(dhi, i);
    alive &= ~(((u64)rhi << 32) | rlo);
  }
  const int npop = (int)__popcll(alive);
  u64 kept = alive;
  if (npop > room) kept = __ballot(((alive >> lane) & 1ull) && (int)__popcll(alive & lt_lane) < room);
  *npop_out = min(npop, room);
  return kept;
}

// wait until block b is published; returns its kept bits, *nk = kept count after it
__device__ __forceinline__ u64 scan_wait_block(const int* nkeptw, const u64* keepw, int b, int* nk) {
  int v;
  u64 kbv;
  for (;;) {
    // count first, bits second: LDS serves a wave's requests in order
    v = __hip_atomic_load(&nkeptw[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
    kbv = __hip_atomic_load(&keepw[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (v >= 0) break;
    __builtin_amdgcn_s_sleep(1);
  }
  *nk = __builtin_amdgcn_readfirstlane(v);
  return rfl64(kbv);
}

__device__ __forceinline__ void scan_publish(int* nkeptw, u64* keepw, int b, u64 kept, int nk, int lane) {
  if (lane == 0) __hip_atomic_store(&keepw[b], kept, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // kept bits are in LDS before the count
  if (lane == 0) __hip_atomic_store(&nkeptw[b], nk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// LDSMAT = true : chunk of at most SCAN_LDS_CAND candidates, Lt holds the packed image (k_nms_mask
//                 packed = 1), staged in LDS up front; output rows prefetched before the walk.
// LDSMAT = false: any chunk size, Lt word-major in global memory, column words streamed SCAN_RING
//                 blocks ahead.
struct ScanParams {          // pointer tables: one entry per image of the batch (blockIdx.y)
  PerImg<NmsState*> st;
  PerImg<const u64*> Lt, diag_up, removed_init;
  PerImg<const float4*> sboxes, sorig;
  PerImg<const uint32_t*> sorted_idx;
  PerImg<int32_t*> out_idx;
  PerImg<float4*> out_boxes, kept_boxes;
  PerImg<int32_t*> out_count, out_done;
  PerImg<float4*> as_rois;                   // optional fused _assign_levels outputs (null = disabled)
  PerImg<int32_t*> as_level;
  PerImg<int64_t*> as_perm;
  PerImg<int32_t*> as_counts;
  PerImg<int32_t*> as_order;                 // optional (K <= 1024): spatial processing order of the assigned RoIs
  float ord_inv_h;
  int n, use_init, K, min_level, max_level;
  int fail_empty;      // last sync-free chunk of a job whose outputs feed further kernels through out_count (the fused
                       // proposal stages): an INCOMPLETE result is reported as zero proposals (+ *out_done = 0), so
                       // that nothing downstream runs on a partial / stale RoI list
};

// the job could not be completed inside its sync-free chunks: empty result (thread 0 of the workgroup)
__device__ __forceinline__ void scan_fail_empty(int32_t* out_count, int32_t* out_done, int32_t* level_counts, int nl) {
  *out_count = 0;
  if (out_done) *out_done = 0;
  if (level_counts)
    for (int L = 0; L < nl; ++L) level_counts[L] = 0;
}

template <bool LDSMAT>
__global__ void __launch_bounds__(SCAN_THREADS) k_nms_scan(ScanParams sp) {
  const int img = blockIdx.y;
  NmsState* st = sp.st.v[img];
  const int n = sp.n, use_init = sp.use_init, K = sp.K;
  const u64* __restrict__ Lt = sp.Lt.v[img];
  const u64* __restrict__ diag_up = sp.diag_up.v[img];
  const u64* __restrict__ removed_init = sp.removed_init.v[img];
  const float4* __restrict__ sboxes = sp.sboxes.v[img];
  const float4* __restrict__ sorig = sp.sorig.v[img];
  const uint32_t* __restrict__ sorted_idx = sp.sorted_idx.v[img];
  int32_t* __restrict__ out_idx = sp.out_idx.v[img];
  float4* out_boxes = sp.out_boxes.v[img];
  float4* __restrict__ kept_boxes = sp.kept_boxes.v[img];
  int32_t* __restrict__ out_count = sp.out_count.v[img];
  int32_t* __restrict__ out_done = sp.out_done.v[img];
  AssignOut ao;
  ao.rois = sp.as_rois.v[img]; ao.level = sp.as_level.v[img]; ao.perm = sp.as_perm.v[img];
  ao.counts = sp.as_counts.v[img]; ao.min_level = sp.min_level; ao.max_level = sp.max_level;
  (void)sboxes;
  // The hand-off words are accessed with relaxed workgroup-scope atomics (plain ds_read / ds_write that
  // the compiler may not cache or hoist).  NOT volatile: volatile __shared__ accesses are lowered to
  // flat sc0 sc1 instructions + vmcnt(0), an order of magnitude slower and they drain the prefetch.
  __shared__ u64 keepw[NMS_WORDS];
  __shared__ int nkeptw[NMS_WORDS];           // < 0: block not published yet
  __shared__ int keptpre[NMS_WORDS + 1];
  __shared__ int lvl_cnt[ODET_MAX_LEVELS][NMS_WORDS];
  __shared__ int lvl_tot[ODET_MAX_LEVELS];
  extern __shared__ __align__(16) u64 mat[];  // packed strictly-lower matrix (LDSMAT)
  if (st->done) return;                       // uniform: an earlier chunk finished the job
  // a later chunk that found nothing left in its order (the ranked selection is exhausted): the state stays
  // "not done" for whoever continues (the per-image fallback on the full order, or the caller's out_done check)
  if (sp.use_init && st->chunk_m == 0 && st->pos < sp.n - st->n_invalid) {
    if (sp.fail_empty && threadIdx.x == 0) scan_fail_empty(out_count, out_done, ao.counts, ao.max_level - ao.min_level + 1);
    return;
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int m = LDSMAT ? min(st->chunk_m, SCAN_LDS_CAND) : st->chunk_m;   // (<= by construction)
  const int pos0 = st->pos;
  const int nk0 = st->kept;
  const int nblk = (m + 63) >> 6;
  const int b_own = SCAN_Q * w;               // first block of this wave
  const int jbase = w * (64 * SCAN_Q) + lane; // candidate of (q, lane) = jbase + 64 q
  const u64 lt_lane = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  // Who writes which candidate's outputs.  The walk needs a wave to own 8 CONSECUTIVE blocks; the output tail does not, and with
  // the LDS-resident chunk (<= 1536 candidates = the first three waves' blocks) that ownership left five of the eight waves idle
  // through the tail (round 6: outputs 5.6 us, level assignment + order 4.5 us of an 18 us launch).  LDSMAT: thread t writes
  // candidates t + 512 q (q < 3) = lane t & 63 of blocks (t >> 6) + 8 q -- every wave, a third of the work each.
  constexpr int TQ = LDSMAT ? SCAN_LDS_CAND / SCAN_THREADS : SCAN_Q;
  static_assert(!LDSMAT || TQ * SCAN_THREADS == SCAN_LDS_CAND, "tail ownership covers the LDS-resident chunk");
  auto tblk = [&](int q) { return LDSMAT ? w + SCAN_WAVES * q : b_own + q; };
  auto tcand = [&](int q) { return LDSMAT ? (int)threadIdx.x + SCAN_THREADS * q : jbase + 64 * q; };

  u64 supm[SCAN_Q];                           // wave-uniform: candidates of own block q suppressed / absent
  u64 dg[SCAN_Q];
  float4 obox[SCAN_Q];
  uint32_t oidx[SCAN_Q];
#pragma unroll
  for (int q = 0; q < SCAN_Q; ++q) {
    const int j = jbase + 64 * q;
    u64 sm = __ballot(j >= m);
    if (use_init && b_own + q < nblk) sm |= removed_init[b_own + q];
    supm[q] = sm;
    dg[q] = (j < m) ? diag_up[j] : 0ull;
  }
  if (threadIdx.x < NMS_WORDS) { keepw[threadIdx.x] = 0ull; nkeptw[threadIdx.x] = -1; }
  // (the tail's level counts: every (level, block) entry is read by the partition's scan, the LDS-resident chunk writes 24 blocks)
  static_assert(ODET_MAX_LEVELS * NMS_WORDS == SCAN_THREADS, "one level-count entry per thread");
  (&lvl_cnt[0][0])[threadIdx.x] = 0;
  int nk = nk0;
  bool stop = false;

  // Block walk as a dataflow between the waves (no workgroup barrier inside): wave w first consumes the
  // blocks of the earlier waves in order as they are published through LDS (kept bits + running kept
  // count), folding Lt[b][j] & kept[b] into the "suppressed" masks of its own blocks, then resolves its
  // own eight blocks back to back -- a hand-off to the next wave only every 512 candidates.  A wave that
  // sees the kept count reach K stops; later waves see the same published count and stop before they
  // would wait for a block that is never published.
  if (LDSMAT) {
    // output rows, fetched now (independent of every decision) and written after the walk
#pragma unroll
    for (int q = 0; q < TQ; ++q) {
      const int j = tcand(q);
      oidx[q] = (j < m) ? sorted_idx[pos0 + j] : 0u;
      obox[q] = (j < m) ? sorig[j] : make_float4(0, 0, 0, 0);
    }
    // stage the packed image: one flat copy, all loads of a thread in flight together
    {
      const int units = SCAN_MAT_OFF(nblk - 1, nblk) / 2;    // 16-byte units (0 when nblk <= 1)
      const uint4* src = reinterpret_cast<const uint4*>(Lt);
      uint4* dst = reinterpret_cast<uint4*>(mat);
      uint4 tmp[SCAN_STAGE_ITEMS];
#pragma unroll
      for (int i = 0; i < SCAN_STAGE_ITEMS; ++i) {
        const int u = i * SCAN_THREADS + threadIdx.x;
        tmp[i] = (u < units) ? src[u] : make_uint4(0, 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < SCAN_STAGE_ITEMS; ++i) {
        const int u = i * SCAN_THREADS + threadIdx.x;
        if (u < units) dst[u] = tmp[i];
      }
    }
    __syncthreads();
    if (b_own < nblk) {
      for (int b = 0; b < b_own; ++b) {
        if (stop) break;
        // words of (source block b -> own blocks); reads past nblk hit the slack, their masks are unused
        const u64* col = mat + SCAN_MAT_OFF(b, nblk) + (b_own - (b + 1)) * 64 + lane;
        u64 wd[SCAN_Q];
#pragma unroll
        for (int q = 0; q < SCAN_Q; ++q) wd[q] = col[q * 64];
        const u64 kb = scan_wait_block(nkeptw, keepw, b, &nk);
#pragma unroll
        for (int q = 0; q < SCAN_Q; ++q) supm[q] |= __ballot((wd[q] & kb) != 0ull);
        if (nk >= K) stop = true;
      }
#pragma unroll
      for (int a = 0; a < SCAN_Q; ++a) {
        const int b = b_own + a;
        if (!stop && b < nblk) {                            // uniform
          const u64* col = mat + SCAN_MAT_OFF(b, nblk) + lane;   // rows of block b+1 first
          u64 wd[SCAN_Q];
#pragma unroll
          for (int q = a + 1; q < SCAN_Q; ++q) wd[q] = col[(q - a - 1) * 64];
          int add;
          const u64 kept = scan_resolve_block(~supm[a], dg[a], K - nk, lane, lt_lane, &add);
          nk += add;
          scan_publish(nkeptw, keepw, b, kept, nk, lane);
#pragma unroll
          for (int q = a + 1; q < SCAN_Q; ++q) supm[q] |= __ballot((wd[q] & kept) != 0ull);
          if (nk >= K) stop = true;
        }
      }
    }
  } else {
    u64 ring[SCAN_RING][SCAN_Q];
    u64 own[SCAN_Q][SCAN_Q];                  // own[a][q] (a < q): Lt[own block a][own candidate of block q]
#pragma unroll
    for (int q = 0; q < SCAN_Q; ++q) {
      const int j = jbase + 64 * q;
#pragma unroll
      for (int k = 0; k < SCAN_RING; ++k)
        ring[k][q] = (j < m && k < b_own) ? Lt[(size_t)k * NMS_CHUNK + j] : 0ull;
    }
#pragma unroll
    for (int a = 0; a < SCAN_Q; ++a) {
#pragma unroll
      for (int q = 0; q < SCAN_Q; ++q) {
        const int j = jbase + 64 * q;
        own[a][q] = (q > a && j < m) ? Lt[(size_t)(b_own + a) * NMS_CHUNK + j] : 0ull;
      }
    }
    __syncthreads();
    if (b_own < nblk) {
      for (int b0 = 0; b0 < b_own; b0 += SCAN_RING) {      // b_own is a multiple of SCAN_RING
#pragma unroll
        for (int k = 0; k < SCAN_RING; ++k) {
          if (!stop) {
            const int b = b0 + k;
            const u64 kb = scan_wait_block(nkeptw, keepw, b, &nk);
#pragma unroll
            for (int q = 0; q < SCAN_Q; ++q) {
              supm[q] |= __ballot((ring[k][q] & kb) != 0ull);
              const int j = jbase + 64 * q;
              ring[k][q] = (j < m && b + SCAN_RING < b_own) ? Lt[(size_t)(b + SCAN_RING) * NMS_CHUNK + j] : 0ull;
            }
            if (nk >= K) stop = true;
          }
        }
      }
#pragma unroll
      for (int a = 0; a < SCAN_Q; ++a) {
        const int b = b_own + a;
        if (!stop && b < nblk) {                            // uniform
          int add;
          const u64 kept = scan_resolve_block(~supm[a], dg[a], K - nk, lane, lt_lane, &add);
          nk += add;
          scan_publish(nkeptw, keepw, b, kept, nk, lane);
#pragma unroll
          for (int q = a + 1; q < SCAN_Q; ++q) supm[q] |= __ballot((own[a][q] & kept) != 0ull);
          if (nk >= K) stop = true;
        }
      }
    }
  }
  __syncthreads();
  // outputs: kept candidates in score order
  if (w == 0) {
    const int pc = (int)__popcll(keepw[lane]);
    const int inc = wave_incl_scan(pc);
    keptpre[lane] = nk0 + inc - pc;
    if (lane == 63) keptpre[NMS_WORDS] = nk0 + inc;
  }
  __syncthreads();
  const int nkf = keptpre[NMS_WORDS];
  const int np = pos0 + m;
  const int done = (nkf >= K || np >= n - st->n_invalid) ? 1 : 0;
  const bool want_assign = ao.rois && done && out_boxes;
  const bool fused_lv = want_assign && nk0 == 0;
  const int nl = ao.max_level - ao.min_level + 1;
  // every block of the chunk is covered by the tail's ownership (rows past m are not kept)
  int myp[SCAN_Q], mylv[SCAN_Q], myrank[SCAN_Q];
  bool iskept[SCAN_Q];
#pragma unroll
  for (int q = 0; q < TQ; ++q) {
    const int c = tcand(q);
    const u64 kbits = keepw[tblk(q)];
    iskept[q] = (c < m) && ((kbits >> lane) & 1ull);
    myp[q] = iskept[q] ? keptpre[tblk(q)] + (int)__popcll(kbits & lt_lane) : -1;
    if (!LDSMAT) {
      oidx[q] = iskept[q] ? sorted_idx[pos0 + c] : 0u;
      obox[q] = iskept[q] ? sorig[c] : make_float4(0, 0, 0, 0);
    }
  }
#pragma unroll
  for (int q = 0; q < TQ; ++q) {
    mylv[q] = -1;
    myrank[q] = 0;
    if (iskept[q]) {
      const int p = myp[q];
      out_idx[p] = (int32_t)oidx[q];
      if (out_boxes) out_boxes[p] = obox[q];
      kept_boxes[p] = d_norm_box(obox[q]);     // == sboxes[c], without a load between the stores
      if (fused_lv) mylv[q] = d_roi_level(obox[q], ao.min_level, ao.max_level);
    }
    if (fused_lv) {
#pragma unroll
      for (int L = 0; L < ODET_MAX_LEVELS; ++L) {
        if (L < nl) {
          const u64 bal = __ballot(mylv[q] == L);
          if (mylv[q] == L) myrank[q] = (int)__popcll(bal & lt_lane);
          if (lane == 0) lvl_cnt[L][tblk(q)] = (int)__popcll(bal);
        }
      }
    }
  }
  __syncthreads();                 // everyone has read the state before it changes
  if (threadIdx.x == 0) {
    st->kept = nkf;
    st->pos = np;
    st->done = done;
    if (sp.fail_empty && !done) {
      scan_fail_empty(out_count, out_done, ao.counts, nl);
    } else {
      *out_count = nkf;
      if (out_done) *out_done = done;
    }
  }
  if (fused_lv) {
    // _assign_levels (base_fpn_model.py:303-324) as a stable partition by level without re-reading
    // anything: per (level, block) counts -> exclusive scan over the 64 blocks (wave L scans level L),
    // then every kept candidate knows its slot.
    if (w < nl) {
      const int v = lvl_cnt[w][lane];
      const int inc = wave_incl_scan(v);
      lvl_cnt[w][lane] = inc - v;
      if (lane == 63) { lvl_tot[w] = inc; ao.counts[w] = inc; }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < TQ; ++q) {
      if (mylv[q] >= 0) {
        int base = 0;
        for (int L = 0; L < mylv[q]; ++L) base += lvl_tot[L];
        const int pos = base + lvl_cnt[mylv[q]][tblk(q)] + myrank[q];
        ao.rois[pos] = obox[q];
        ao.level[pos] = mylv[q];
        ao.perm[pos] = myp[q];
        myrank[q] = pos;                       // (from here on: the RoI's row in the assigned list, for the order below)
      }
    }
  } else if (want_assign) {
    // several chunks contributed (fallback path): partition the complete RoI list from memory
    __shared__ int lds_al[ODET_MAX_LEVELS * 17];
    __threadfence_block();
    __syncthreads();               // out_boxes rows written above are visible to the whole workgroup
    d_assign_levels_block<SCAN_THREADS>(out_boxes, nkf, ao.min_level, ao.max_level, ao.rois, ao.level, ao.perm,
                                        ao.counts, lds_al);
  }
  // Spatial processing order of the RoIs for the RoI kernel, in this launch instead of one of its own (odet_roi_order):
  // the assigned RoIs were written by this workgroup just above.  A counting sort by (level, y band of 1/32 of the
  // image) -- 256 buckets: what the order is for is that the RoIs in flight at a time tap one band of one pyramid
  // level, and a band already holds fewer RoIs than are in flight; inside a bucket the order is whatever the LDS
  // atomics give (results do not depend on the processing order).  ~1 us instead of the ~8 us of a full sort.
  // Without an assignment (job not finished yet / reported empty) the order is the identity, so that the RoI kernel
  // visits -- and zero-fills -- every row.
  int32_t* __restrict__ order = sp.as_order.v[img];
  if (order && fused_lv) {
    // (round 6) the common case -- one chunk finished the job: every kept candidate still holds its box, its level and its row
    // of the assigned list in registers, so the counting sort runs on those instead of re-reading ao.rois / ao.level from
    // memory behind the stores above (a store -> load round trip through L2 in the middle of a one-workgroup launch)
    __syncthreads();                               // (lvl_cnt / lvl_tot were read above; they become the buckets now)
    const int tid = threadIdx.x;
    int* ocnt = &lvl_cnt[0][0];                    // [256]
    int* obase = ocnt + 256;                       // [256]
    for (int i = tid; i < 256; i += SCAN_THREADS) ocnt[i] = 0;
    __syncthreads();
    int bktq[SCAN_Q], slotq[SCAN_Q];
#pragma unroll
    for (int q = 0; q < TQ; ++q) {
      bktq[q] = -1;
      if (mylv[q] >= 0) {
        const float4 bx = obox[q];
        const int l = min(max(mylv[q], 0), 7);
        const int qy = min(max((int)((bx.y + bx.w) * 0.5f * sp.ord_inv_h * 4096.0f), 0), 4095);
        bktq[q] = l * 32 + (qy >> 7);
        slotq[q] = atomicAdd(&ocnt[bktq[q]], 1);
      }
    }
    for (int r = nkf + tid; r < K; r += SCAN_THREADS) order[r] = r;     // padded rows stay behind the valid ones
    __syncthreads();
    {
      int total;
      const int v = (tid < 256) ? ocnt[tid] : 0;
      const int ex = block_excl_scan(v, keptpre, &total);
      if (tid < 256) obase[tid] = ex;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < TQ; ++q)
      if (bktq[q] >= 0) order[obase[bktq[q]] + slotq[q]] = myrank[q];
  } else if (order) {
    __threadfence_block();
    __syncthreads();
    const int tid = threadIdx.x;
    // (single-level models: no level assignment, the order is taken over the kept RoIs themselves, level 0)
    const float4* __restrict__ osrc = want_assign ? ao.rois : ((done && !ao.rois) ? out_boxes : nullptr);
    const int32_t* __restrict__ olvl = want_assign ? ao.level : nullptr;
    if (osrc) {
      int* ocnt = &lvl_cnt[0][0];                  // [256] (the level assignment is done with it)
      int* obase = ocnt + 256;                     // [256]
      for (int i = tid; i < 256; i += SCAN_THREADS) ocnt[i] = 0;
      __syncthreads();
      int bkt[2], slot[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int r = tid + e * SCAN_THREADS;
        bkt[e] = -1;
        if (r < nkf) {
          const float4 bx = osrc[r];
          const int l = olvl ? min(max(olvl[r], 0), 7) : 0;
          const int qy = min(max((int)((bx.y + bx.w) * 0.5f * sp.ord_inv_h * 4096.0f), 0), 4095);
          bkt[e] = l * 32 + (qy >> 7);
          slot[e] = atomicAdd(&ocnt[bkt[e]], 1);
        } else if (r < K) {
          order[r] = r;                            // padded rows stay behind the valid ones
        }
      }
      __syncthreads();
      {
        int total;
        const int v = (tid < 256) ? ocnt[tid] : 0;
        const int ex = block_excl_scan(v, keptpre, &total);
        if (tid < 256) obase[tid] = ex;
      }
      __syncthreads();
#pragma unroll
      for (int e = 0; e < 2; ++e)
        if (bkt[e] >= 0) order[obase[bkt[e]] + slot[e]] = tid + e * SCAN_THREADS;
    } else {
      if (tid < K) order[tid] = tid;
      if (tid + SCAN_THREADS < K) order[tid + SCAN_THREADS] = tid + SCAN_THREADS;
    }
  }
}

// ------------------------------------------------------------------------------ host side --
struct NmsWorkspace {
  NmsHeader* hdr;
  uint32_t *keys_a, *vals_a, *keys_b, *vals_b, *hist;
  u64* cand;
  float4 *sboxes, *sorig, *kept_boxes;
  u64 *Lt, *diag, *removed_init;
};

static size_t nms_carve(int n, int max_out, void* ws, size_t ws_bytes, NmsWorkspace* o) {
  OdetArena ar{(char*)ws, ws ? ws_bytes : (size_t)-1, 0};
  size_t nn = (size_t)(n > 0 ? n : 1);
  size_t kk = (size_t)(max_out > 0 ? max_out : 1);
#define TAKE(field, T, count)                                 \
  do {                                                        \
    T* p_ = ar.take<T>(count);                                \
    if (o) o->field = ws ? p_ : nullptr;                      \
  } while (0)
  TAKE(hdr, NmsHeader, 1);
  TAKE(keys_a, uint32_t, nn);
  TAKE(vals_a, uint32_t, nn);
  TAKE(keys_b, uint32_t, nn);
  TAKE(vals_b, uint32_t, std::max(nn, (size_t)NMS_SEL_MAX));
  TAKE(hist, uint32_t, odet_sort_hist_entries(n));
  TAKE(cand, u64, NMS_SEL_MAX);
  TAKE(sboxes, float4, NMS_CHUNK);
  TAKE(sorig, float4, NMS_CHUNK);
  TAKE(kept_boxes, float4, kk);
  TAKE(Lt, u64, (size_t)NMS_WORDS * NMS_CHUNK);
  TAKE(diag, u64, NMS_CHUNK);
  TAKE(removed_init, u64, NMS_WORDS);
#undef TAKE
  return ar.off + 256;
}

extern "C" size_t odet_nms_workspace_bytes(int n, int max_output) {
  return nms_carve(n, max_output, nullptr, 0, nullptr);
}

// Candidates selected for the first chunk: with few overlaps K kept boxes need barely more than K
// candidates, so the first bit matrix is sized ~1.5 K instead of 4096 (7x fewer IoU tiles at K = 1000).
static int first_chunk_target(int n, int K, int first_chunk) {
  long long c = ((long long)K * 3 / 2 + 63) / 64 * 64;
  if (c > 64) c -= 32;   // half a block of slack for boundary ties: K = 1000 -> 1504 -> at most 24 blocks
  // caller's choice (odet_fpn_step_t.nms_first_chunk): a WIDER first chunk -- up to NMS_CHUNK candidates on the
  // register-ring scan -- lets score distributions with heavy suppression finish in the one sync-free chunk that
  // batched launches have (measured: trained-like clustered scores 20.7k img/s batched against 10.8k through
  // the per-image two-chunk path; costs 12 % when the narrow chunk would have done)
  if (first_chunk > 0) c = std::max<long long>(c, ((long long)first_chunk + 63) / 64 * 64 - 32);
  if (c < 256) c = 256;
  if (c > NMS_CHUNK) c = NMS_CHUNK;
  if (c > n) c = n;
  return (int)c;
}

static inline int tri_tiles(int cap) {
  int nb = (cap + 63) / 64;
  return nb * (nb + 1) / 2;
}

struct NmsImage {           // per-image pointers of a job
  const float4* boxes_in;   // PREP_NMS: boxes; PREP_RP: anchors
  const float* deltas;
  const float* scores;
  const float2* logits;
  float4* boxes_out;        // PREP_RP / PREP_FPN / PREP_FRCNN: decoded boxes (carved from the workspace)
  const float4* boxes;      // boxes the NMS runs on
  int32_t* out_idx;
  float* out_boxes;
  int32_t* out_count;
  int32_t* out_done;
  AssignOut assign;
  void* ws;                 // NMS workspace of this image
  size_t ws_bytes;
};

struct NmsJob {
  int mode;                 // PREP_*
  PrepParams prep;          // common parameters (means, stds, clip, anchor tables); pointer tables filled by nms_run
  int n, K;
  float thr;
  int blind_chunks;
  int ws_clean;             // caller promise: the workspace header is clean (see NmsHeader): no k_zero_headers launch
  int fail_empty;           // sync-free jobs: report an incomplete result as zero proposals (ScanParams::fail_empty)
  int first_chunk;          // 0 = auto (~1.5 K candidates), else candidates of the first chunk (<= NMS_CHUNK)
  int B;                    // images in the batch (1..ODET_MAX_BATCH)
  NmsImage img[ODET_MAX_BATCH];
};

template <typename T, typename F>
static PerImg<T> per_img(const NmsJob& J, F get) {
  PerImg<T> t;
  for (int i = 0; i < ODET_MAX_BATCH; ++i) t.v[i] = get(i < J.B ? i : 0);
  return t;
}

// Chunk 0 (select path) is always enqueued.  blind_chunks > 1: the fallback (full sort + further
// chunks) is enqueued without looking at the device state; its kernels exit at once when chunk 0
// finished the job.  out_done == nullptr: exact mode -- afterwards the host reads the state (one
// sync per further chunk) until the device reports done.  out_done != nullptr: sync-free mode --
// exactly blind_chunks chunks, *out_done tells the caller whether the result is complete.
// Batches (B > 1) run chunk 0 of every image in the same launches (blockIdx.y = image); the
// fallback is per image, so batches require the sync-free mode with blind_chunks == 1.
static int nms_run(NmsJob& J, hipStream_t st) {
  const int n = J.n, K = J.K, B = J.B;
  if (B < 1 || B > ODET_MAX_BATCH) return odet_set_error(ODET_E_INVALID, "odet_nms: batch %d out of range", B);
  NmsWorkspace w[ODET_MAX_BATCH];
  const size_t need = nms_carve(n, K, nullptr, 0, nullptr);
  for (int i = 0; i < B; ++i) {
    if (!J.img[i].ws || J.img[i].ws_bytes < need)
      return odet_set_error(ODET_E_WORKSPACE, "odet_nms: workspace too small (%zu < %zu)", J.img[i].ws_bytes, need);
    nms_carve(n, K, J.img[i].ws, J.img[i].ws_bytes, &w[i]);
  }
  if (B > 1) {
    for (int i = 0; i < B; ++i)
      if (!J.img[i].out_done)
        return odet_set_error(ODET_E_INVALID, "odet_nms: batches need the sync-free mode (out_done)");
  }
  static OdetPerDeviceOnce once;    // (executor threads may arrive here together)
  ODET_HIP(once.run([] {
    return hipFuncSetAttribute((const void*)k_nms_scan<true>, hipFuncAttributeMaxDynamicSharedMemorySize, SCAN_DYN_LDS);
  }));
  const PerImg<NmsHeader*> hdrs = per_img<NmsHeader*>(J, [&](int i) { return w[i].hdr; });
  const PerImg<const uint32_t*> keys = per_img<const uint32_t*>(J, [&](int i) { return (const uint32_t*)w[i].keys_a; });
  if (!J.ws_clean) {
    hipLaunchKernelGGL(k_zero_headers, dim3((unsigned)((sizeof(NmsHeader) / 16 + 255) / 256), B), dim3(256), 0, st, hdrs);
    ODET_LAUNCH_CHECK();
  }
  // 1. prepare
  J.prep.n = n;
  J.prep.boxes_in = per_img<const float4*>(J, [&](int i) { return J.img[i].boxes_in; });
  J.prep.deltas = per_img<const float*>(J, [&](int i) { return J.img[i].deltas; });
  J.prep.scores = per_img<const float*>(J, [&](int i) { return J.img[i].scores; });
  J.prep.logits = per_img<const float2*>(J, [&](int i) { return J.img[i].logits; });
  J.prep.boxes_out = per_img<float4*>(J, [&](int i) { return J.img[i].boxes_out; });
  J.prep.keys = per_img<uint32_t*>(J, [&](int i) { return w[i].keys_a; });
  J.prep.hdr = hdrs;
  {
    dim3 grid((n + PREP_TILE - 1) / PREP_TILE, B), block(256);
    if (J.mode == PREP_NMS)
      hipLaunchKernelGGL(HIP_KERNEL_NAME(k_rp_prepare<PREP_NMS>), grid, block, 0, st, J.prep);
    else if (J.mode == PREP_RP)
      hipLaunchKernelGGL(HIP_KERNEL_NAME(k_rp_prepare<PREP_RP>), grid, block, 0, st, J.prep);
    else if (J.mode == PREP_FRCNN)
      hipLaunchKernelGGL(HIP_KERNEL_NAME(k_rp_prepare<PREP_FRCNN>), grid, block, 0, st, J.prep);
    else
      hipLaunchKernelGGL(HIP_KERNEL_NAME(k_rp_prepare<PREP_FPN>), grid, block, 0, st, J.prep);
    ODET_LAUNCH_CHECK();
  }
  // 2. select + order the first chunk
  const uint32_t target = (uint32_t)first_chunk_target(n, K, J.first_chunk);
  // chunk 0 runs on the LDS-resident scan when its candidates are guaranteed to fit 24 blocks
  const bool lds0 = target <= (uint32_t)SCAN_LDS_CAND;
  const uint32_t limit = lds0 ? (uint32_t)SCAN_LDS_CAND : (uint32_t)NMS_CHUNK;      // capacity of the chunk-0 scan
  // Sync-free jobs that ask for a second chunk get it from the SAME radix selection: one NMS_CHUNK more candidates
  // are selected and ranked, and chunk 1 runs on that order in launches shared by the whole batch -- no per-image
  // radix sort of all anchors for it.
  const int blind_req = J.blind_chunks < 1 ? 1 : J.blind_chunks;
  const bool wide = J.img[0].out_done != nullptr && blind_req >= 2;
  const uint32_t sel_target = wide ? (uint32_t)std::min<long long>(n, (long long)target + NMS_CHUNK) : target;
  const uint32_t sel_limit = wide ? (uint32_t)NMS_SEL_MAX : limit;
  const PerImg<const float4*> nboxes = per_img<const float4*>(J, [&](int i) { return J.img[i].boxes; });
  const PerImg<float4*> sboxes = per_img<float4*>(J, [&](int i) { return w[i].sboxes; });
  const PerImg<float4*> sorig = per_img<float4*>(J, [&](int i) { return w[i].sorig; });
  {
    dim3 grid((n + SEL_TILE - 1) / SEL_TILE, B), block(SEL_BLOCK);
    hipLaunchKernelGGL(k_sel_hist2, grid, block, 0, st, hdrs, keys, n, sel_target);
    ODET_LAUNCH_CHECK();
    const PerImg<u64*> cand = per_img<u64*>(J, [&](int i) { return w[i].cand; });
    hipLaunchKernelGGL(k_sel_compact, grid, block, 0, st, hdrs, keys, n, sel_target, sel_limit, cand);
    ODET_LAUNCH_CHECK();
    if (wide) {
      // a boundary bin too large for the selection (thousands of equal keys: a saturated RPN) is split exactly in
      // (key, index) order; both launches exit at once when the bin fitted.  Scratch: the bit-matrix buffer (idle here).
      // (per block 256 counters of 2 bytes: fits the 2 MiB of the matrix buffer up to ~8.4 M anchors)
      ODET_REQUIRE((size_t)grid.x * 256 * sizeof(unsigned short) <= (size_t)NMS_WORDS * NMS_CHUNK * sizeof(u64),
                   "odet_nms: %d anchors exceed the tie split's scratch (the bit-matrix buffer)", n);
      const PerImg<unsigned short*> bh = per_img<unsigned short*>(J, [&](int i) { return (unsigned short*)w[i].Lt; });
      hipLaunchKernelGGL(k_sel_tie_hist, grid, block, 0, st, hdrs, keys, n, bh);
      ODET_LAUNCH_CHECK();
      // (the split hands over what the selection was asked for -- first chunk + one more --, not the list's capacity: ranking
      // by counting is quadratic in the candidates, 79 -> 37 us per 8 images)
      hipLaunchKernelGGL(k_sel_tie_take, grid, block, 0, st, hdrs, keys, n, sel_target,
                         per_img<const unsigned short*>(J, [&](int i) { return (const unsigned short*)w[i].Lt; }), cand);
      ODET_LAUNCH_CHECK();
    }
    const int rank_wgs = (std::min(n, (int)sel_limit) + 63) / 64;
    hipLaunchKernelGGL(k_sel_rank, dim3(rank_wgs, B), dim3(RANK_THREADS), 0, st, hdrs, n, (int)limit,
                       per_img<const u64*>(J, [&](int i) { return (const u64*)w[i].cand; }), J.prep, J.mode,
                       per_img<uint32_t*>(J, [&](int i) { return w[i].vals_b; }), sboxes, sorig);
    ODET_LAUNCH_CHECK();
  }
  // 3./4. chunk 0
  const PerImg<const NmsState*> cstates = per_img<const NmsState*>(J, [&](int i) { return (const NmsState*)&w[i].hdr->st; });
  const PerImg<const float4*> csboxes = per_img<const float4*>(J, [&](int i) { return (const float4*)w[i].sboxes; });
  const PerImg<u64*> Lts = per_img<u64*>(J, [&](int i) { return w[i].Lt; });
  const PerImg<u64*> diags = per_img<u64*>(J, [&](int i) { return w[i].diag; });
  ScanParams sp;
  sp.st = per_img<NmsState*>(J, [&](int i) { return &w[i].hdr->st; });
  sp.Lt = per_img<const u64*>(J, [&](int i) { return (const u64*)w[i].Lt; });
  sp.diag_up = per_img<const u64*>(J, [&](int i) { return (const u64*)w[i].diag; });
  sp.removed_init = per_img<const u64*>(J, [&](int i) { return (const u64*)w[i].removed_init; });
  sp.sboxes = csboxes;
  sp.sorig = per_img<const float4*>(J, [&](int i) { return (const float4*)w[i].sorig; });
  sp.sorted_idx = per_img<const uint32_t*>(J, [&](int i) { return (const uint32_t*)w[i].vals_b; });
  sp.out_idx = per_img<int32_t*>(J, [&](int i) { return J.img[i].out_idx; });
  sp.out_boxes = per_img<float4*>(J, [&](int i) { return (float4*)J.img[i].out_boxes; });
  sp.kept_boxes = per_img<float4*>(J, [&](int i) { return w[i].kept_boxes; });
  sp.out_count = per_img<int32_t*>(J, [&](int i) { return J.img[i].out_count; });
  sp.out_done = per_img<int32_t*>(J, [&](int i) { return J.img[i].out_done; });
  sp.as_rois = per_img<float4*>(J, [&](int i) { return J.img[i].assign.rois; });
  sp.as_level = per_img<int32_t*>(J, [&](int i) { return J.img[i].assign.level; });
  sp.as_perm = per_img<int64_t*>(J, [&](int i) { return J.img[i].assign.perm; });
  sp.as_counts = per_img<int32_t*>(J, [&](int i) { return J.img[i].assign.counts; });
  sp.as_order = per_img<int32_t*>(J, [&](int i) { return K <= ODET_FUSED_ORDER_MAX_ROIS ? J.img[i].assign.order : nullptr; });
  sp.ord_inv_h = 1.0f / (J.prep.hmax + 1.0f);
  sp.n = n; sp.use_init = 0; sp.K = K;
  const int blind_n = J.blind_chunks < 1 ? 1 : J.blind_chunks;
  const bool fail_empty = J.fail_empty && J.img[0].out_done != nullptr;
  sp.fail_empty = (fail_empty && blind_n == 1) ? 1 : 0;         // (set on the LAST sync-free chunk only)
  sp.min_level = J.img[0].assign.min_level; sp.max_level = J.img[0].assign.max_level;
  {
    const int cap0 = std::min((int)limit, (n + 63) / 64 * 64);
    hipLaunchKernelGGL(k_nms_mask, dim3(tri_tiles(cap0), B), dim3(256), 0, st, cstates, csboxes, J.thr, Lts, diags,
                       lds0 ? 1 : 0);
    ODET_LAUNCH_CHECK();
    if (lds0)
      hipLaunchKernelGGL(HIP_KERNEL_NAME(k_nms_scan<true>), dim3(1, B), dim3(SCAN_THREADS), SCAN_DYN_LDS, st, sp);
    else
      hipLaunchKernelGGL(HIP_KERNEL_NAME(k_nms_scan<false>), dim3(1, B), dim3(SCAN_THREADS), 0, st, sp);
    ODET_LAUNCH_CHECK();
  }
  int blind = J.blind_chunks < 1 ? 1 : J.blind_chunks;
  // 5. fallback of image i: full order, then chunks of 4096 from wherever chunk 0 stopped.  Every launch is
  //    guarded by the image's device-side `done` word.  Batches (sync-free by construction) run exactly
  //    blind - 1 further chunks per image, one image after the other: ~13 launches per image that exit at once
  //    where chunk 0 has already finished -- the price of never asking the host.
  auto fallback = [&](int i, bool host_checks, int chunks) -> int {
    NmsState* state = &w[i].hdr->st;
    const NmsWorkspace& wi = w[i];
    auto shift = [&](auto tbl) { auto t = tbl; for (int k = 0; k < ODET_MAX_BATCH; ++k) t.v[k] = tbl.v[i]; return t; };
    if (host_checks && blind == 1) {
      NmsState h;
      ODET_HIP(hipMemcpyAsync(&h, state, sizeof(h), hipMemcpyDeviceToHost, st));
      ODET_HIP(hipStreamSynchronize(st));
      if (h.done) return ODET_OK;
    }
    uint32_t* sorted = nullptr;
    int rc = odet_sort_keys_desc(n, wi.keys_a, wi.vals_a, wi.keys_b, wi.vals_b, wi.hist, &state->done, &sorted, st);
    if (rc != ODET_OK) return rc;
    ScanParams si = sp;
    si.st = shift(sp.st); si.Lt = shift(sp.Lt); si.diag_up = shift(sp.diag_up); si.removed_init = shift(sp.removed_init);
    si.sboxes = shift(sp.sboxes); si.sorig = shift(sp.sorig); si.out_idx = shift(sp.out_idx);
    si.out_boxes = shift(sp.out_boxes); si.kept_boxes = shift(sp.kept_boxes); si.out_count = shift(sp.out_count);
    si.out_done = shift(sp.out_done); si.as_rois = shift(sp.as_rois); si.as_level = shift(sp.as_level);
    si.as_perm = shift(sp.as_perm); si.as_counts = shift(sp.as_counts); si.as_order = shift(sp.as_order);
    si.use_init = 1;
    si.fail_empty = 0;
    for (int k = 0; k < ODET_MAX_BATCH; ++k) si.sorted_idx.v[k] = (const uint32_t*)sorted;
    const PerImg<const NmsState*> cst = shift(cstates);
    const PerImg<const float4*> csb = shift(csboxes);
    const PerImg<u64*> lt = shift(Lts), dg = shift(diags);
    const int max_chunks = host_checks ? (n + NMS_CHUNK - 1) / NMS_CHUNK + 1 : chunks;
    PrepParams prep_i = J.prep;                 // image i at entry 0 (grid.y == 1)
    prep_i.boxes_in = shift(J.prep.boxes_in); prep_i.deltas = shift(J.prep.deltas);
    PerImg<NmsState*> st_i;
    PerImg<const float4*> boxes_i, kept_i;
    PerImg<const uint32_t*> sorted_i;
    PerImg<float4*> sb_i, so_i;
    PerImg<u64*> ri_i;
    for (int k = 0; k < ODET_MAX_BATCH; ++k) {
      st_i.v[k] = state; boxes_i.v[k] = J.img[i].boxes; sorted_i.v[k] = sorted; sb_i.v[k] = wi.sboxes;
      so_i.v[k] = wi.sorig; kept_i.v[k] = wi.kept_boxes; ri_i.v[k] = wi.removed_init;
    }
    for (int c = 1; c <= max_chunks; ++c) {
      if (host_checks && c >= blind && !(c == 1 && blind == 1)) {   // (c == 1 && blind == 1: just seen "not done")
        NmsState h;
        ODET_HIP(hipMemcpyAsync(&h, state, sizeof(h), hipMemcpyDeviceToHost, st));
        ODET_HIP(hipStreamSynchronize(st));
        if (h.done) break;
      }
      const int cap = std::min(NMS_CHUNK, (n + 63) / 64 * 64);
      hipLaunchKernelGGL(k_nms_gather, dim3((cap + 255) / 256, 1), dim3(256), 0, st, st_i, n, cap, 0, prep_i, J.mode,
                         sorted_i, sb_i, so_i);
      ODET_LAUNCH_CHECK();
      hipLaunchKernelGGL(k_nms_cross, dim3((cap + 255) / 256, 1), dim3(256), 0, st, cst, csb, kept_i, J.thr, ri_i);
      ODET_LAUNCH_CHECK();
      hipLaunchKernelGGL(k_nms_mask, dim3(tri_tiles(cap), 1), dim3(256), 0, st, cst, csb, J.thr, lt, dg, 0);
      ODET_LAUNCH_CHECK();
      si.fail_empty = (!host_checks && fail_empty && c == max_chunks) ? 1 : 0;
      hipLaunchKernelGGL(HIP_KERNEL_NAME(k_nms_scan<false>), dim3(1, 1), dim3(SCAN_THREADS), 0, st, si);
      ODET_LAUNCH_CHECK();
    }
    return ODET_OK;
  };
  const bool sync_free = J.img[0].out_done != nullptr;    // (batches: checked above for every image)
  if (sync_free) {
    if (blind == 1) return ODET_OK;              // one chunk: the caller checks *out_done
    // chunk 1 of every image from the ranked selection, in launches shared by the batch
    {
      ScanParams s1 = sp;
      s1.use_init = 1;                           // (sorted_idx stays the ranked selection)
      s1.fail_empty = (fail_empty && blind == 2) ? 1 : 0;
      const int cap = std::min(NMS_CHUNK, (n + 63) / 64 * 64);
      hipLaunchKernelGGL(k_nms_gather, dim3((cap + 255) / 256, B), dim3(256), 0, st, sp.st, n, cap, 1, J.prep, J.mode,
                         sp.sorted_idx, sboxes, sorig);
      ODET_LAUNCH_CHECK();
      hipLaunchKernelGGL(k_nms_cross, dim3((cap + 255) / 256, B), dim3(256), 0, st, cstates, csboxes,
                         per_img<const float4*>(J, [&](int i) { return (const float4*)w[i].kept_boxes; }), J.thr,
                         per_img<u64*>(J, [&](int i) { return w[i].removed_init; }));
      ODET_LAUNCH_CHECK();
      hipLaunchKernelGGL(k_nms_mask, dim3(tri_tiles(cap), B), dim3(256), 0, st, cstates, csboxes, J.thr, Lts, diags, 0);
      ODET_LAUNCH_CHECK();
      hipLaunchKernelGGL(HIP_KERNEL_NAME(k_nms_scan<false>), dim3(1, B), dim3(SCAN_THREADS), 0, st, s1);
      ODET_LAUNCH_CHECK();
    }
    // chunks 2..: the full order of every image (radix sort of all n keys) and further chunks of NMS_CHUNK candidates,
    // all in launches shared by the batch (blockIdx.y = image; an image that is done skips every one of them): with
    // heavily overlapping proposals (a random-init or a real RPN: 5-6 candidates per kept box) every image of a batch
    // comes here, and one image after the other this was 13 launches = 140 us PER IMAGE.
    if (blind > 2) {
      OdetSortImage si_[ODET_MAX_BATCH];
      for (int i = 0; i < B; ++i)
        si_[i] = OdetSortImage{w[i].keys_a, w[i].vals_a, w[i].keys_b, w[i].vals_b, w[i].hist, &w[i].hdr->st.done};
      const int rc = odet_sort_keys_desc_batch(n, B, si_, st);
      if (rc != ODET_OK) return rc;
      ScanParams s2 = sp;
      s2.use_init = 1;
      s2.sorted_idx = per_img<const uint32_t*>(J, [&](int i) { return (const uint32_t*)w[i].vals_a; });
      const PerImg<const float4*> kept = per_img<const float4*>(J, [&](int i) { return (const float4*)w[i].kept_boxes; });
      const PerImg<u64*> rinit = per_img<u64*>(J, [&](int i) { return w[i].removed_init; });
      const int cap = std::min(NMS_CHUNK, (n + 63) / 64 * 64);
      for (int c = 2; c < blind; ++c) {
        hipLaunchKernelGGL(k_nms_gather, dim3((cap + 255) / 256, B), dim3(256), 0, st, sp.st, n, cap, 0, J.prep, J.mode,
                           s2.sorted_idx, sboxes, sorig);
        ODET_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_nms_cross, dim3((cap + 255) / 256, B), dim3(256), 0, st, cstates, csboxes, kept, J.thr, rinit);
        ODET_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_nms_mask, dim3(tri_tiles(cap), B), dim3(256), 0, st, cstates, csboxes, J.thr, Lts, diags, 0);
        ODET_LAUNCH_CHECK();
        s2.fail_empty = (fail_empty && c == blind - 1) ? 1 : 0;
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_nms_scan<false>), dim3(1, B), dim3(SCAN_THREADS), 0, st, s2);
        ODET_LAUNCH_CHECK();
      }
    }
    return ODET_OK;
  }
  return fallback(0, true, 0);                   // exact mode (single image): the host follows the chunks
}

static int nms_trivial(int32_t* out_count, int32_t* out_done, hipStream_t st) {
  ODET_HIP(hipMemsetAsync(out_count, 0, sizeof(int32_t), st));
  if (out_done) {
    ODET_HIP(hipMemsetAsync(out_done, 0, sizeof(int32_t), st));
    ODET_HIP(hipMemsetAsync(out_done, 1, 1, st));   // little-endian int32 1
  }
  return ODET_OK;
}

static void no_assign(AssignOut* a) {
  a->rois = nullptr; a->level = nullptr; a->perm = nullptr; a->counts = nullptr; a->order = nullptr;
  a->min_level = 0; a->max_level = 0;
}

static void job_init(NmsJob* J, int mode, int n, int K, float thr, int blind_chunks, int B) {
  memset(J, 0, sizeof(*J));
  J->mode = mode; J->n = n; J->K = K; J->thr = thr; J->blind_chunks = blind_chunks; J->B = B;
  // the fused proposal stages feed further kernels through out_count: incomplete sync-free results are reported empty
  J->fail_empty = (mode == PREP_FPN || mode == PREP_FRCNN) ? 1 : 0;
  for (int i = 0; i < ODET_MAX_BATCH; ++i) no_assign(&J->img[i].assign);
}

extern "C" int odet_nms(const float* boxes, const float* scores, int n, int max_output, float iou_threshold,
                        int32_t* out_idx, float* out_boxes, int32_t* out_count, int blind_chunks,
                        int32_t* out_done, void* workspace, size_t workspace_bytes, odet_stream_t stream) {
  ODET_REQUIRE(n >= 0 && max_output >= 0, "odet_nms: negative size");
  ODET_REQUIRE(out_count, "odet_nms: null out_count");
  if (n == 0 || max_output == 0) return nms_trivial(out_count, out_done, (hipStream_t)stream);
  ODET_REQUIRE(boxes && scores && out_idx, "odet_nms: null pointer");
  NmsJob J;
  job_init(&J, PREP_NMS, n, max_output, iou_threshold, blind_chunks, 1);
  NmsImage& im = J.img[0];
  im.boxes_in = (const float4*)boxes;
  im.scores = scores;
  im.boxes = (const float4*)boxes;
  im.out_idx = out_idx; im.out_boxes = out_boxes; im.out_count = out_count; im.out_done = out_done;
  im.ws = workspace; im.ws_bytes = workspace_bytes;
  return nms_run(J, (hipStream_t)stream);
}

static size_t rp_workspace_bytes(int n, int max_output) {
  return odet_nms_workspace_bytes(n, max_output) + odet_align_up((size_t)(n > 0 ? n : 1) * 16, 256) +
         odet_align_up((size_t)(max_output > 0 ? max_output : 1) * sizeof(int32_t), 256) + 512;
}

extern "C" size_t odet_region_proposal_workspace_bytes(int n, int max_output) {
  return rp_workspace_bytes(n, max_output);
}

// carve: [NMS workspace (header first)] [boxes n] [idx K]; fills the image's workspace / box fields
static int rp_carve(int n, int K, void* workspace, size_t workspace_bytes, const char* who, NmsImage* im,
                    int32_t* out_idx) {
  size_t need = rp_workspace_bytes(n, K);
  if (!workspace || workspace_bytes < need)
    return odet_set_error(ODET_E_WORKSPACE, "%s: workspace too small (%zu < %zu)", who, workspace_bytes, need);
  size_t nb = odet_align_up(odet_nms_workspace_bytes(n, K), 256);
  OdetArena ar{(char*)workspace + nb, workspace_bytes - nb, 0};
  im->boxes_out = ar.take<float4>((size_t)n);
  int32_t* idx_buf = ar.take<int32_t>((size_t)K);
  im->boxes = im->boxes_out;
  im->out_idx = out_idx ? out_idx : idx_buf;
  im->ws = workspace;
  im->ws_bytes = nb;
  return ODET_OK;
}

extern "C" int odet_region_proposal(const float* deltas, const float* anchors, const float* scores, int n,
                                    int image_h, int image_w, const float* means, const float* stds,
                                    int max_output, float iou_threshold, float* out_rois, int32_t* out_idx,
                                    int32_t* out_count, int blind_chunks, int32_t* out_done, void* workspace,
                                    size_t workspace_bytes, odet_stream_t stream) {
  ODET_REQUIRE(n >= 0 && max_output >= 0, "odet_region_proposal: negative size");
  ODET_REQUIRE(out_count, "odet_region_proposal: null out_count");
  ODET_REQUIRE(image_h > 0 && image_w > 0, "odet_region_proposal: bad image shape");
  if (n == 0 || max_output == 0) return nms_trivial(out_count, out_done, (hipStream_t)stream);
  ODET_REQUIRE(deltas && anchors && scores && means && stds && out_rois, "odet_region_proposal: null pointer");
  NmsJob J;
  job_init(&J, PREP_RP, n, max_output, iou_threshold, blind_chunks, 1);
  NmsImage& im = J.img[0];
  int rc = rp_carve(n, max_output, workspace, workspace_bytes, "odet_region_proposal", &im, out_idx);
  if (rc != ODET_OK) return rc;
  im.boxes_in = (const float4*)anchors;
  im.deltas = deltas;
  im.scores = scores;
  for (int k = 0; k < 4; ++k) { J.prep.means.v[k] = means[k]; J.prep.stds.v[k] = stds[k]; }
  J.prep.wmax = (float)(image_w - 1); J.prep.hmax = (float)(image_h - 1);
  im.out_boxes = out_rois; im.out_count = out_count; im.out_done = out_done;
  return nms_run(J, (hipStream_t)stream);
}

extern "C" size_t odet_fpn_proposals_workspace_bytes(int n, int max_output) {
  return rp_workspace_bytes(n, max_output);
}

// shared by odet_fpn_proposals (B = 1) and odet_fpn_step_enqueue_batch (B images in the same launches)
int odet_fpn_proposals_batch(const FpnProposalIO* io, int B, int num_levels, int A, const int* fh, const int* fw,
                             const int* stride, const float* wh, int image_h, int image_w, const float* means,
                             const float* stds, int max_output, float iou_threshold, int min_level, int max_level,
                             int blind_chunks, hipStream_t st, int first_chunk, int ws_clean) {
  ODET_REQUIRE(io && fh && fw && stride && wh && means && stds, "odet_fpn_proposals: null pointer");
  ODET_REQUIRE(first_chunk >= 0 && first_chunk <= NMS_CHUNK, "odet_fpn_proposals: nms_first_chunk %d out of range (0..%d)", first_chunk, NMS_CHUNK);
  ODET_REQUIRE(B >= 1 && B <= ODET_MAX_BATCH, "odet_fpn_proposals: batch %d out of range", B);
  ODET_REQUIRE(num_levels > 0 && num_levels <= ODET_MAX_LEVELS, "odet_fpn_proposals: num_levels %d out of range",
               num_levels);
  ODET_REQUIRE(A > 0 && A <= ODET_MAX_ANCHORS_PER_CELL, "odet_fpn_proposals: A %d out of range", A);
  ODET_REQUIRE(image_h > 0 && image_w > 0 && max_output > 0, "odet_fpn_proposals: bad sizes");
  NmsJob J;
  job_init(&J, PREP_FPN, 0, max_output, iou_threshold, blind_chunks, B);
  J.first_chunk = first_chunk;
  J.ws_clean = ws_clean ? 1 : 0;
  FpnAnchorParams& p = J.prep.fpn;
  p.num_levels = num_levels;
  p.A = A;
  int64_t total = 0;
  for (int l = 0; l < num_levels; ++l) {
    ODET_REQUIRE(fh[l] >= 0 && fw[l] > 0 && stride[l] > 0, "odet_fpn_proposals: bad level %d", l);
    p.fw[l] = fw[l];
    p.stride[l] = stride[l];
    p.start[l] = (int)total;
    total += (int64_t)fh[l] * fw[l] * A;
    ODET_REQUIRE(total < (1ll << 31), "odet_fpn_proposals: too many anchors");
  }
  for (int l = num_levels; l <= ODET_MAX_LEVELS; ++l) p.start[l] = (int)total;
  for (int i = 0; i < num_levels * A * 2; ++i) p.wh[i] = wh[i];
  const int n = (int)total;
  J.n = n;
  for (int k = 0; k < 4; ++k) { J.prep.means.v[k] = means[k]; J.prep.stds.v[k] = stds[k]; }
  J.prep.wmax = (float)(image_w - 1); J.prep.hmax = (float)(image_h - 1);
  for (int i = 0; i < B; ++i) {
    const FpnProposalIO& a = io[i];
    ODET_REQUIRE(a.rpn_logits && a.rpn_deltas && a.out_rois && a.out_count, "odet_fpn_proposals: null pointer");
    const bool assign = a.out_sorted_rois != nullptr;
    if (assign) {
      ODET_REQUIRE(a.out_level && a.out_perm && a.out_level_counts, "odet_fpn_proposals: null level outputs");
      ODET_REQUIRE(max_level >= min_level && max_level - min_level < ODET_MAX_LEVELS, "odet_fpn_proposals: bad levels");
      if (max_output > ODET_ASSIGN_MAX_ROIS)
        return odet_set_error(ODET_E_LIMIT, "odet_fpn_proposals: max_output %d exceeds %d", max_output,
                              ODET_ASSIGN_MAX_ROIS);
    }
    if (n == 0) {
      if (assign) ODET_HIP(hipMemsetAsync(a.out_level_counts, 0, sizeof(int32_t) * (max_level - min_level + 1), st));
      int rc0 = nms_trivial(a.out_count, a.out_done, st);
      if (rc0 != ODET_OK) return rc0;
      continue;
    }
    NmsImage& im = J.img[i];
    int rc = rp_carve(n, max_output, a.workspace, a.workspace_bytes, "odet_fpn_proposals", &im, a.out_idx);
    if (rc != ODET_OK) return rc;
    im.logits = (const float2*)a.rpn_logits;
    im.deltas = a.rpn_deltas;
    im.out_boxes = a.out_rois; im.out_count = a.out_count; im.out_done = a.out_done;
    if (assign) {
      im.assign.rois = (float4*)a.out_sorted_rois; im.assign.level = a.out_level; im.assign.perm = a.out_perm;
      im.assign.counts = a.out_level_counts;
      im.assign.order = a.out_order;
    }
    im.assign.min_level = min_level; im.assign.max_level = max_level;
  }
  if (n == 0) return ODET_OK;
  return nms_run(J, st);
}

extern "C" int odet_fpn_proposals(const float* rpn_logits, const float* rpn_deltas, int num_levels, int A,
                                  const int* fh, const int* fw, const int* stride, const float* wh, int image_h,
                                  int image_w, const float* means, const float* stds, int max_output,
                                  float iou_threshold, int min_level, int max_level, float* out_rois,
                                  int32_t* out_idx, int32_t* out_count, float* out_sorted_rois, int32_t* out_level,
                                  int64_t* out_perm, int32_t* out_level_counts, int32_t* out_order, int blind_chunks,
                                  int32_t* out_done, void* workspace, size_t workspace_bytes, odet_stream_t stream) {
  ODET_REQUIRE(!out_order || (out_sorted_rois && max_output <= ODET_FUSED_ORDER_MAX_ROIS),
               "odet_fpn_proposals: out_order needs out_sorted_rois and max_output <= %d (use odet_roi_order beyond)",
               ODET_FUSED_ORDER_MAX_ROIS);
  FpnProposalIO io{rpn_logits, rpn_deltas, out_rois, out_idx, out_count, out_sorted_rois, out_level, out_perm,
                   out_level_counts, out_done, workspace, workspace_bytes, out_order};
  return odet_fpn_proposals_batch(&io, 1, num_levels, A, fh, fw, stride, wh, image_h, image_w, means, stds,
                                  max_output, iou_threshold, min_level, max_level, blind_chunks, (hipStream_t)stream, 0);
}

extern "C" size_t odet_frcnn_proposals_workspace_bytes(int n, int max_output) {
  return rp_workspace_bytes(n, max_output);
}

// shared by odet_frcnn_proposals (B = 1) and the single-level step descriptors (B images in the same launches)
int odet_frcnn_proposals_batch(const FpnProposalIO* io, int B, const float* anchor_base, int A, int feat_stride, int fh,
                               int fw, int image_h, int image_w, const float* means, const float* stds, int max_output,
                               float iou_threshold, int blind_chunks, hipStream_t st, int first_chunk, int ws_clean) {
  ODET_REQUIRE(io && anchor_base && means && stds, "odet_frcnn_proposals: null pointer");
  ODET_REQUIRE(B >= 1 && B <= ODET_MAX_BATCH, "odet_frcnn_proposals: batch %d out of range", B);
  ODET_REQUIRE(first_chunk >= 0 && first_chunk <= NMS_CHUNK, "odet_frcnn_proposals: nms_first_chunk %d out of range (0..%d)",
               first_chunk, NMS_CHUNK);
  ODET_REQUIRE(A > 0 && A <= ODET_MAX_ANCHORS_PER_CELL, "odet_frcnn_proposals: A %d out of range", A);
  ODET_REQUIRE(feat_stride > 0 && fh >= 0 && fw >= 0 && image_h > 0 && image_w > 0 && max_output > 0,
               "odet_frcnn_proposals: bad sizes");
  const int64_t total = (int64_t)fh * fw * A;
  ODET_REQUIRE(total < (1ll << 31), "odet_frcnn_proposals: too many anchors");
  const int n = (int)total;
  NmsJob J;
  job_init(&J, PREP_FRCNN, n, max_output, iou_threshold, blind_chunks, B);
  J.first_chunk = first_chunk;
  J.ws_clean = ws_clean ? 1 : 0;
  J.prep.fpn.A = A;
  J.prep.fpn.fw[0] = fw;
  J.prep.fpn.stride[0] = feat_stride;
  for (int i = 0; i < A * 4; ++i) J.prep.fpn.wh[i] = anchor_base[i];
  for (int k = 0; k < 4; ++k) { J.prep.means.v[k] = means[k]; J.prep.stds.v[k] = stds[k]; }
  J.prep.wmax = (float)(image_w - 1); J.prep.hmax = (float)(image_h - 1);
  for (int i = 0; i < B; ++i) {
    const FpnProposalIO& a = io[i];
    ODET_REQUIRE(a.rpn_logits && a.rpn_deltas && a.out_rois && a.out_count, "odet_frcnn_proposals: null pointer");
    ODET_REQUIRE(!a.out_order || max_output <= ODET_FUSED_ORDER_MAX_ROIS,
                 "odet_frcnn_proposals: out_order needs max_output <= %d", ODET_FUSED_ORDER_MAX_ROIS);
    if (n == 0) {
      int rc0 = nms_trivial(a.out_count, a.out_done, st);
      if (rc0 != ODET_OK) return rc0;
      continue;
    }
    NmsImage& im = J.img[i];
    int rc = rp_carve(n, max_output, a.workspace, a.workspace_bytes, "odet_frcnn_proposals", &im, a.out_idx);
    if (rc != ODET_OK) return rc;
    im.logits = (const float2*)a.rpn_logits;
    im.deltas = a.rpn_deltas;
    im.out_boxes = a.out_rois; im.out_count = a.out_count; im.out_done = a.out_done;
    im.assign.order = a.out_order;             // (no level assignment: the order is taken over the kept RoIs themselves)
  }
  if (n == 0) return ODET_OK;
  return nms_run(J, st);
}

extern "C" int odet_frcnn_proposals(const float* rpn_logits, const float* rpn_deltas, const float* anchor_base, int A,
                                    int feat_stride, int fh, int fw, int image_h, int image_w, const float* means,
                                    const float* stds, int max_output, float iou_threshold, float* out_rois,
                                    int32_t* out_idx, int32_t* out_count, int blind_chunks, int32_t* out_done,
                                    void* workspace, size_t workspace_bytes, odet_stream_t stream) {
  ODET_REQUIRE(rpn_logits && rpn_deltas && out_rois && out_count, "odet_frcnn_proposals: null pointer");
  FpnProposalIO io{rpn_logits, rpn_deltas, out_rois, out_idx, out_count, nullptr, nullptr, nullptr, nullptr, out_done,
                   workspace, workspace_bytes, nullptr};
  return odet_frcnn_proposals_batch(&io, 1, anchor_base, A, feat_stride, fh, fw, image_h, image_w, means, stds, max_output,
                                    iou_threshold, blind_chunks, (hipStream_t)stream, 0, 0);
}

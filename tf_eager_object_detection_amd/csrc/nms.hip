// Exact greedy NMS (tf.image.non_max_suppression / NonMaxSuppressionV3 semantics) and the
// fused RegionProposal path (model/region_proposal.py:55-81).
//
// TF's kernel is a serial loop: pop the best remaining candidate, test it against every box
// kept so far, keep it if no IoU > threshold, stop at max_output.  The result only depends on
// the candidates in descending-score order until max_output are kept, so the GPU version is:
//
//   1. stable radix sort of all n (score, index) pairs                       (sort.hip)
//   2. per chunk of sorted candidates (first chunk ~1.5*K, later ones 4096), until K kept or
//      the candidates are exhausted:
//      a. k_nms_gather : boxes of the chunk in sorted order, corners normalised
//      b. k_nms_cross  : (chunks > 0) candidates vs. boxes kept by earlier chunks -> bitmask
//      c. k_nms_mask   : upper-triangular suppression bit matrix of the chunk, one wave per
//                        64x64 tile (whole chip busy), 64-bit word per (row, column block)
//      d. k_nms_scan   : ONE workgroup walks the chunk in 64-candidate blocks: wave 0 resolves a
//                        block serially on the scalar unit (ctz over the alive mask, diagonal
//                        rows via v_readlane), then four waves OR the kept rows (prefetched
//                        one block ahead into registers) into the removed-bit vector in LDS.
//                        Outputs are written by all threads after the walk.
//
// The decision for every candidate is the same predicate TF evaluates (d_iou_gt), so kept
// indices are identical to the serial algorithm, including the early stop at max_output.
#include "odet_internal.h"

#define NMS_CHUNK 4096
#define NMS_WORDS (NMS_CHUNK / 64)   // 64 u64 words per mask row

struct NmsState {      // zeroed by one memset per call
  int32_t n_invalid;   // scores TF would not push into its heap (NaN, <= lowest float)
  int32_t kept;        // boxes kept so far
  int32_t pos;         // sorted candidates consumed so far
  int32_t done;        // kept == K or pos == n_valid
  int32_t chunk_m;     // size of the current chunk
  int32_t pad[3];
};

// a. gather the chunk's boxes in sorted order (corner-normalised) --------------------------
__global__ void __launch_bounds__(256) k_nms_gather(NmsState* st, int n, int cap, const float4* __restrict__ boxes,
                                                    const uint32_t* __restrict__ sorted_idx,
                                                    float4* __restrict__ sboxes) {
  const int pos = st->pos, nv = n - st->n_invalid;
  const int m = st->done ? 0 : min(cap, nv - pos);
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i == 0) st->chunk_m = m;
  if (i < m) sboxes[i] = d_norm_box(boxes[sorted_idx[pos + i]]);
}

// b. chunk candidates vs boxes kept by earlier chunks ---------------------------------------
__global__ void __launch_bounds__(256) k_nms_cross(const NmsState* st, const float4* __restrict__ sboxes,
                                                   const float4* __restrict__ kept_boxes, float thr,
                                                   unsigned long long* __restrict__ removed_init) {
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
  unsigned long long bal = __ballot(sup && j < m);
  if ((threadIdx.x & 63) == 0) removed_init[j >> 6] = bal;
}

// c. suppression bit matrix -------------------------------------------------------------------
// grid (col block, row block), one wave per 64x64 tile, only col block >= row block.
// mask[row][cb] bit j: candidate cb*64+j (later in score order) is suppressed by `row`.
__global__ void __launch_bounds__(64) k_nms_mask(const NmsState* st, const float4* __restrict__ sboxes, float thr,
                                                 unsigned long long* __restrict__ mask,
                                                 unsigned long long* __restrict__ diag) {
  const int cb = blockIdx.x, rb = blockIdx.y;
  if (cb < rb) return;
  const int m = st->chunk_m;
  if (rb * 64 >= m) return;
  __shared__ float4 cbox[64];
  __shared__ float carea[64];
  const int lane = threadIdx.x;
  const int col = cb * 64 + lane;
  const float4 c = (col < m) ? sboxes[col] : make_float4(0, 0, 0, 0);   // zero area never suppresses
  cbox[lane] = c;
  carea[lane] = d_box_area(c);
  __syncthreads();
  const int row = rb * 64 + lane;
  const float4 r = (row < m) ? sboxes[row] : make_float4(0, 0, 0, 0);
  const float ra = d_box_area(r);
  unsigned long long bits = 0;
#pragma unroll 8
  for (int j = 0; j < 64; ++j) {
    bool s = d_iou_gt(r, ra, cbox[j], carea[j], thr);
    bits |= s ? (1ull << j) : 0ull;
  }
  if (cb == rb) {
    // only later candidates (j > lane) can be suppressed by this row
    const unsigned long long later = (lane == 63) ? 0ull : (~0ull << (lane + 1));
    bits &= later;
    if (row < m) diag[row] = bits;
  }
  if (row < m) mask[(size_t)row * NMS_WORDS + cb] = bits;
}

// d. serial scan ------------------------------------------------------------------------------
// LDS: two staging buffers of one 64-candidate block each (64 rows x 64 words + 64 diagonal words).
// Iteration b: (1) issue the global loads of block b+1 (they fly during the resolve), (2) wave 0
// resolves block b from LDS only, (3) all waves OR the kept rows of block b (LDS) into `removed`,
// (4) the loaded rows of block b+1 are written to the other staging buffer.
#define SCAN_THREADS 256
#define SCAN_WAVES 4
#define SCAN_ROWS 16   // rows of a 64-candidate block owned by each wave
#define SCAN_STAGE_WORDS (64 * NMS_WORDS + 64)
#define SCAN_DYN_LDS (2 * SCAN_STAGE_WORDS * 8)

__device__ __forceinline__ unsigned long long rfl64(unsigned long long v) {
  uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
  uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return ((unsigned long long)hi << 32) | lo;
}

__global__ void __launch_bounds__(SCAN_THREADS) k_nms_scan(
    NmsState* st, int n, const unsigned long long* __restrict__ mask, const unsigned long long* __restrict__ diag,
    const unsigned long long* __restrict__ removed_init, int use_init, const float4* __restrict__ sboxes,
    const uint32_t* __restrict__ sorted_idx, const float4* __restrict__ boxes, int K,
    int32_t* __restrict__ out_idx, float4* __restrict__ out_boxes, float4* __restrict__ kept_boxes,
    int32_t* __restrict__ out_count, int32_t* __restrict__ out_done) {
  extern __shared__ __align__(16) unsigned long long stage[];   // [2][SCAN_STAGE_WORDS]
  __shared__ unsigned long long removed[NMS_WORDS];
  __shared__ unsigned long long keptbits[NMS_WORDS];
  __shared__ int keptpre[NMS_WORDS + 1];
  __shared__ unsigned long long s_kept64;
  __shared__ int s_nkept;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int m = st->chunk_m;
  const int pos0 = st->pos;
  const int nk0 = st->kept;
  const int nblk = (m + 63) >> 6;
  if (threadIdx.x < NMS_WORDS) {
    unsigned long long r = use_init ? removed_init[threadIdx.x] : 0ull;
    // candidates beyond m do not exist
    const int lo = threadIdx.x * 64;
    if (lo + 64 > m) r |= (lo >= m) ? ~0ull : (~0ull << (m - lo));
    removed[threadIdx.x] = r;
    keptbits[threadIdx.x] = 0ull;
  }
  if (threadIdx.x == 0) { s_nkept = nk0; s_kept64 = 0; }

  unsigned long long nrows[SCAN_ROWS];
  unsigned long long nd = 0;
  // block 0 -> staging buffer 0
  if (nblk > 0) {
#pragma unroll
    for (int q = 0; q < SCAN_ROWS; ++q) {
      const int r = w * SCAN_ROWS + q;
      stage[(size_t)r * NMS_WORDS + lane] = (r < m) ? mask[(size_t)r * NMS_WORDS + lane] : 0ull;
    }
    if (w == 0) stage[64 * NMS_WORDS + lane] = (lane < m) ? diag[lane] : 0ull;
  }
  __syncthreads();

  const unsigned long long lt_lane = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  for (int b = 0; b < nblk; ++b) {
    unsigned long long* cur = stage + (size_t)(b & 1) * SCAN_STAGE_WORDS;
    unsigned long long* nxt = stage + (size_t)((b + 1) & 1) * SCAN_STAGE_WORDS;
    const bool more = (b + 1 < nblk);
    // (1) loads of the next block (independent of any decision)
    if (more) {
#pragma unroll
      for (int q = 0; q < SCAN_ROWS; ++q) {
        const int r = (b + 1) * 64 + w * SCAN_ROWS + q;
        nrows[q] = (r < m) ? mask[(size_t)r * NMS_WORDS + lane] : 0ull;
      }
      if (w == 0) { const int r = (b + 1) * 64 + lane; nd = (r < m) ? diag[r] : 0ull; }
    }
    // (2) resolve this block: only candidates that actually suppress a still-alive later candidate
    //     need a serial step; everything else that is alive is kept.
    if (w == 0) {
      unsigned long long alive = ~rfl64(removed[b]);
      const int cnt0 = __builtin_amdgcn_readfirstlane(s_nkept);
      const unsigned long long d = cur[64 * NMS_WORDS + lane];
      const uint32_t dlo = (uint32_t)d, dhi = (uint32_t)(d >> 32);
      for (;;) {
        const bool is_s = ((alive >> lane) & 1ull) && ((d & alive) != 0ull);
        const unsigned long long supp = __ballot(is_s);
        if (supp == 0) break;
        const int i = __builtin_ctzll(supp);
        const uint32_t rlo = __builtin_amdgcn_readlane(dlo, i);
        const uint32_t rhi = __builtin_amdgcn_readlane(dhi, i);
        alive &= ~(((unsigned long long)rhi << 32) | rlo);
      }
      // stop at max_output: keep only the first (K - cnt0) survivors
      const int room = K - cnt0;
      const int npop = (int)__popcll(alive);
      unsigned long long kept = alive;
      if (npop > room)
        kept = __ballot(((alive >> lane) & 1ull) && (int)__popcll(alive & lt_lane) < room);
      if (lane == 0) { s_kept64 = kept; s_nkept = cnt0 + min(npop, room); keptbits[b] = kept; }
    }
    __syncthreads();
    // (3) OR the kept rows into the removed vector
    const unsigned long long kept = rfl64(s_kept64);
    const int nk = __builtin_amdgcn_readfirstlane(s_nkept);
    unsigned long long acc = 0;
#pragma unroll
    for (int q = 0; q < SCAN_ROWS; ++q) {
      const int r = w * SCAN_ROWS + q;
      if ((kept >> r) & 1ull) acc |= cur[(size_t)r * NMS_WORDS + lane];
    }
    if (lane > b && acc) atomicOr(&removed[lane], acc);
    // (4) stage the next block
    if (more) {
#pragma unroll
      for (int q = 0; q < SCAN_ROWS; ++q) nxt[(size_t)(w * SCAN_ROWS + q) * NMS_WORDS + lane] = nrows[q];
      if (w == 0) nxt[64 * NMS_WORDS + lane] = nd;
    }
    __syncthreads();
    if (nk >= K) break;
  }

  // outputs: kept candidates in score order
  if (threadIdx.x == 0) {
    int run = nk0;
    for (int k = 0; k < NMS_WORDS; ++k) { keptpre[k] = run; run += (int)__popcll(keptbits[k]); }
    keptpre[NMS_WORDS] = run;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < m; c += SCAN_THREADS) {
    const unsigned long long kb = keptbits[c >> 6];
    const int bit = c & 63;
    if ((kb >> bit) & 1ull) {
      const unsigned long long lt = (bit == 0) ? 0ull : (~0ull >> (64 - bit));
      const int p = keptpre[c >> 6] + (int)__popcll(kb & lt);
      const uint32_t oi = sorted_idx[pos0 + c];
      out_idx[p] = (int32_t)oi;
      if (out_boxes) out_boxes[p] = boxes[oi];
      kept_boxes[p] = sboxes[c];
    }
  }
  if (threadIdx.x == 0) {
    const int nk = keptpre[NMS_WORDS];
    const int np = pos0 + m;
    const int done = (nk >= K || np >= n - st->n_invalid) ? 1 : 0;
    st->kept = nk;
    st->pos = np;
    st->done = done;
    *out_count = nk;
    if (out_done) *out_done = done;
  }
}

// ------------------------------------------------------------------------------ host side --
struct NmsWorkspace {
  uint32_t *keys_a, *vals_a, *keys_b, *vals_b, *hist;
  NmsState* state;
  float4 *sboxes, *kept_boxes;
  unsigned long long *mask, *diag, *removed_init;
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
  TAKE(state, NmsState, 1);
  TAKE(keys_a, uint32_t, nn);
  TAKE(vals_a, uint32_t, nn);
  TAKE(keys_b, uint32_t, nn);
  TAKE(vals_b, uint32_t, nn);
  TAKE(hist, uint32_t, odet_sort_hist_entries(n));
  TAKE(sboxes, float4, NMS_CHUNK);
  TAKE(kept_boxes, float4, kk);
  TAKE(mask, unsigned long long, (size_t)NMS_CHUNK * NMS_WORDS);
  TAKE(diag, unsigned long long, NMS_CHUNK);
  TAKE(removed_init, unsigned long long, NMS_WORDS);
#undef TAKE
  return ar.off + 256;
}

extern "C" size_t odet_nms_workspace_bytes(int n, int max_output) {
  return nms_carve(n, max_output, nullptr, 0, nullptr);
}

// Size of the first chunk: with few overlaps K kept boxes need barely more than K candidates, so
// the first bit matrix is sized ~1.5 K instead of 4096 (7x fewer IoU tiles for K = 1000).
static int first_chunk_cap(int n, int K) {
  long long c = ((long long)K * 3 / 2 + 63) / 64 * 64;
  if (c < 256) c = 256;
  if (c > NMS_CHUNK) c = NMS_CHUNK;
  if (c > ((long long)n + 63) / 64 * 64) c = ((long long)n + 63) / 64 * 64;
  return (int)c;
}

// blind_chunks >= 1 chunks are enqueued without looking at the device state (a chunk whose
// predecessor already finished exits at once).  out_done == nullptr: exact mode -- afterwards the
// host reads the state (one sync per further chunk) until the device reports done.
// out_done != nullptr: sync-free mode -- exactly blind_chunks chunks, *out_done tells the caller
// whether the result is complete.
static int nms_run(const float* boxes, const float* scores, int n, int K, float thr, int32_t* out_idx,
                   float* out_boxes, int32_t* out_count, int blind_chunks, int32_t* out_done, void* ws,
                   size_t ws_bytes, hipStream_t st) {
  NmsWorkspace w;
  size_t need = nms_carve(n, K, nullptr, 0, nullptr);
  if (!ws || ws_bytes < need)
    return odet_set_error(ODET_E_WORKSPACE, "odet_nms: workspace too small (%zu < %zu)", ws_bytes, need);
  nms_carve(n, K, ws, ws_bytes, &w);
  static bool attr_set = false;
  if (!attr_set) {
    ODET_HIP(hipFuncSetAttribute((const void*)k_nms_scan, hipFuncAttributeMaxDynamicSharedMemorySize, SCAN_DYN_LDS));
    attr_set = true;
  }
  ODET_HIP(hipMemsetAsync(w.state, 0, sizeof(NmsState), st));
  uint32_t* sorted = nullptr;
  int rc = odet_sort_pairs_desc(scores, n, w.keys_a, w.vals_a, w.keys_b, w.vals_b, w.hist, &w.state->n_invalid,
                                &sorted, st);
  if (rc != ODET_OK) return rc;
  if (blind_chunks < 1) blind_chunks = 1;
  int consumed = 0;   // upper bound of candidates handed to chunks so far
  for (int c = 0; consumed < n; ++c) {
    if (c >= blind_chunks) {
      if (out_done) break;
      // exact mode: need the device's verdict to know whether another chunk is required
      NmsState h;
      ODET_HIP(hipMemcpyAsync(&h, w.state, sizeof(h), hipMemcpyDeviceToHost, st));
      ODET_HIP(hipStreamSynchronize(st));
      if (h.done) break;
    }
    const int cap = (c == 0) ? first_chunk_cap(n, K) : std::min(NMS_CHUNK, (n - consumed + 63) / 64 * 64);
    const int nb = (cap + 63) / 64;
    hipLaunchKernelGGL(k_nms_gather, dim3((cap + 255) / 256), dim3(256), 0, st, w.state, n, cap,
                       (const float4*)boxes, sorted, w.sboxes);
    ODET_LAUNCH_CHECK();
    if (c > 0) {
      hipLaunchKernelGGL(k_nms_cross, dim3((cap + 255) / 256), dim3(256), 0, st, w.state, w.sboxes, w.kept_boxes,
                         thr, w.removed_init);
      ODET_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_nms_mask, dim3(nb, nb), dim3(64), 0, st, w.state, w.sboxes, thr, w.mask, w.diag);
    ODET_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_nms_scan, dim3(1), dim3(SCAN_THREADS), SCAN_DYN_LDS, st, w.state, n, w.mask, w.diag,
                       w.removed_init, c > 0 ? 1 : 0, w.sboxes, sorted, (const float4*)boxes, K, out_idx,
                       (float4*)out_boxes, w.kept_boxes, out_count, out_done);
    ODET_LAUNCH_CHECK();
    consumed += cap;
  }
  return ODET_OK;
}

static int nms_trivial(int32_t* out_count, int32_t* out_done, hipStream_t st) {
  ODET_HIP(hipMemsetAsync(out_count, 0, sizeof(int32_t), st));
  if (out_done) {
    ODET_HIP(hipMemsetAsync(out_done, 0, sizeof(int32_t), st));
    ODET_HIP(hipMemsetAsync(out_done, 1, 1, st));   // little-endian int32 1
  }
  return ODET_OK;
}

extern "C" int odet_nms(const float* boxes, const float* scores, int n, int max_output, float iou_threshold,
                        int32_t* out_idx, float* out_boxes, int32_t* out_count, int blind_chunks,
                        int32_t* out_done, void* workspace, size_t workspace_bytes, odet_stream_t stream) {
  ODET_REQUIRE(n >= 0 && max_output >= 0, "odet_nms: negative size");
  ODET_REQUIRE(out_count, "odet_nms: null out_count");
  if (n == 0 || max_output == 0) return nms_trivial(out_count, out_done, (hipStream_t)stream);
  ODET_REQUIRE(boxes && scores && out_idx, "odet_nms: null pointer");
  return nms_run(boxes, scores, n, max_output, iou_threshold, out_idx, out_boxes, out_count, blind_chunks, out_done,
                 workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" size_t odet_region_proposal_workspace_bytes(int n, int max_output) {
  return odet_align_up((size_t)(n > 0 ? n : 1) * 16, 256) +
         odet_align_up((size_t)(max_output > 0 ? max_output : 1) * sizeof(int32_t), 256) +
         odet_nms_workspace_bytes(n, max_output) + 512;
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
  size_t need = odet_region_proposal_workspace_bytes(n, max_output);
  if (!workspace || workspace_bytes < need)
    return odet_set_error(ODET_E_WORKSPACE, "odet_region_proposal: workspace too small (%zu < %zu)",
                          workspace_bytes, need);
  OdetArena ar{(char*)workspace, workspace_bytes, 0};
  float* boxes = ar.take<float>((size_t)n * 4);
  int32_t* idx_buf = out_idx;
  if (!idx_buf) idx_buf = ar.take<int32_t>((size_t)max_output);
  size_t off = odet_align_up(ar.off, 256);
  // region_proposal.py:59 decode + :63 clip (min_edge=None), fused
  int rc = odet_decode(anchors, deltas, 4, n, means, stds, image_h, image_w, boxes, stream);
  if (rc != ODET_OK) return rc;
  // region_proposal.py:73-76 NMS over all n, :81 gather
  return nms_run(boxes, scores, n, max_output, iou_threshold, idx_buf, out_rois, out_count, blind_chunks, out_done,
                 (char*)workspace + off, workspace_bytes - off, (hipStream_t)stream);
}

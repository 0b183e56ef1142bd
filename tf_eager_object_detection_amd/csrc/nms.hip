// Exact greedy NMS (tf.image.non_max_suppression / NonMaxSuppressionV3 semantics) and the
// fused RegionProposal path (model/region_proposal.py:55-81).
//
// TF's kernel is a serial loop: pop the best remaining candidate, test it against every box
// kept so far, keep it if no IoU > threshold, stop at max_output.  The result only depends on
// the candidates in descending-score order until max_output are kept, so the GPU version is:
//
//   1. stable radix sort of all n (score, index) pairs                       (sort.hip)
//   2. per chunk of NMS_CHUNK = 4096 sorted candidates, until K kept or exhausted:
//      a. k_nms_gather : boxes of the chunk in sorted order, corners normalised
//      b. k_nms_cross  : (chunks > 0) candidates vs. boxes kept by earlier chunks -> bitmask
//      c. k_nms_mask   : upper-triangular 4096x4096 suppression bit matrix, one wave per
//                        64x64 tile (whole chip busy), 64-bit word per (row, column block)
//      d. k_nms_scan   : ONE workgroup walks the chunk in 64-candidate blocks: wave 0 resolves a
//                        block serially on the scalar unit (ctz over the alive mask, diagonal
//                        rows via v_readlane), then four waves OR the kept rows (prefetched
//                        one block ahead into registers) into the removed-bit vector in LDS.
//
// The decision for every candidate is the same predicate TF evaluates (d_iou_gt), so kept
// indices are identical to the serial algorithm, including the early stop at max_output.
#include "odet_internal.h"

#define NMS_CHUNK 4096
#define NMS_WORDS (NMS_CHUNK / 64)   // 64 u64 words per mask row

struct NmsState {
  int32_t n_valid;   // candidates TF would push into its heap
  int32_t kept;      // boxes kept so far
  int32_t pos;       // sorted candidates consumed so far
  int32_t done;      // kept == K or pos == n_valid
  int32_t chunk_m;   // size of the current chunk
  int32_t pad[3];
};

__global__ void k_nms_init(NmsState* st, int32_t* out_count) {
  st->kept = 0;
  st->pos = 0;
  st->done = (st->n_valid == 0) ? 1 : 0;
  st->chunk_m = 0;
  *out_count = 0;
}

// a. gather the chunk's boxes in sorted order (corner-normalised) --------------------------
__global__ void __launch_bounds__(256) k_nms_gather(NmsState* st, const float4* __restrict__ boxes,
                                                    const uint32_t* __restrict__ sorted_idx,
                                                    float4* __restrict__ sboxes) {
  int pos = st->pos, nv = st->n_valid;
  int m = st->done ? 0 : min(NMS_CHUNK, nv - pos);
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i == 0) st->chunk_m = m;
  if (i < m) sboxes[i] = d_norm_box(boxes[sorted_idx[pos + i]]);
}

// b. chunk candidates vs boxes kept by earlier chunks ---------------------------------------
__global__ void __launch_bounds__(256) k_nms_cross(const NmsState* st, const float4* __restrict__ sboxes,
                                                   const float4* __restrict__ kept_boxes, float thr,
                                                   unsigned long long* __restrict__ removed_init) {
  __shared__ float4 kb[256];
  __shared__ float ka[256];
  int m = st->chunk_m, nk = st->kept;
  int j = blockIdx.x * 256 + threadIdx.x;
  float4 b = (j < m) ? sboxes[j] : make_float4(0, 0, 0, 0);
  float area = d_box_area(b);
  bool sup = false;
  for (int k0 = 0; k0 < nk; k0 += 256) {
    __syncthreads();
    if (k0 + threadIdx.x < nk) {
      float4 t = kept_boxes[k0 + threadIdx.x];
      kb[threadIdx.x] = t;
      ka[threadIdx.x] = d_box_area(t);
    }
    __syncthreads();
    int lim = min(256, nk - k0);
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
  int cb = blockIdx.x, rb = blockIdx.y;
  if (cb < rb) return;
  int m = st->chunk_m;
  if (rb * 64 >= m) return;
  __shared__ float4 cbox[64];
  __shared__ float carea[64];
  int lane = threadIdx.x;
  int col = cb * 64 + lane;
  float4 c = (col < m) ? sboxes[col] : make_float4(0, 0, 0, 0);   // zero area never suppresses
  cbox[lane] = c;
  carea[lane] = d_box_area(c);
  __syncthreads();
  int row = rb * 64 + lane;
  float4 r = (row < m) ? sboxes[row] : make_float4(0, 0, 0, 0);
  float ra = d_box_area(r);
  unsigned long long bits = 0;
  int jstart = (cb == rb) ? 0 : 0;
#pragma unroll 8
  for (int j = jstart; j < 64; ++j) {
    bool s = d_iou_gt(r, ra, cbox[j], carea[j], thr);
    bits |= s ? (1ull << j) : 0ull;
  }
  if (cb == rb) {
    // only later candidates (j > lane) can be suppressed by this row
    unsigned long long later = (lane == 63) ? 0ull : (~0ull << (lane + 1));
    bits &= later;
    if (row < m) diag[row] = bits;
  }
  if (row < m) mask[(size_t)row * NMS_WORDS + cb] = bits;
}

// d. serial scan ------------------------------------------------------------------------------
#define SCAN_THREADS 256
#define SCAN_WAVES 4
#define SCAN_ROWS 16   // rows of a 64-candidate block owned by each wave

__device__ __forceinline__ unsigned long long rfl64(unsigned long long v) {
  uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
  uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return ((unsigned long long)hi << 32) | lo;
}

__global__ void __launch_bounds__(SCAN_THREADS) k_nms_scan(
    NmsState* st, const unsigned long long* __restrict__ mask, const unsigned long long* __restrict__ diag,
    const unsigned long long* __restrict__ removed_init, int use_init, const float4* __restrict__ sboxes,
    const uint32_t* __restrict__ sorted_idx, const float4* __restrict__ boxes, int K,
    int32_t* __restrict__ out_idx, float4* __restrict__ out_boxes, float4* __restrict__ kept_boxes,
    int32_t* __restrict__ out_count) {
  __shared__ unsigned long long removed[NMS_WORDS];
  __shared__ unsigned long long s_kept64;
  __shared__ int s_nkept;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int m = st->chunk_m;
  const int pos0 = st->pos;
  const int nblk = (m + 63) >> 6;
  if (threadIdx.x < NMS_WORDS) {
    unsigned long long r = use_init ? removed_init[threadIdx.x] : 0ull;
    // candidates beyond m do not exist
    int lo = threadIdx.x * 64;
    if (lo + 64 > m) r |= (lo >= m) ? ~0ull : (~0ull << (m - lo));
    removed[threadIdx.x] = r;
  }
  if (threadIdx.x == 0) { s_nkept = st->kept; s_kept64 = 0; }
  __syncthreads();

  unsigned long long rows[SCAN_ROWS], nrows[SCAN_ROWS];
  unsigned long long d = 0, nd = 0;
  if (nblk > 0) {
#pragma unroll
    for (int q = 0; q < SCAN_ROWS; ++q) {
      int r = w * SCAN_ROWS + q;
      rows[q] = (r < m) ? mask[(size_t)r * NMS_WORDS + lane] : 0ull;
    }
    if (w == 0) d = (lane < m) ? diag[lane] : 0ull;
  }

  for (int b = 0; b < nblk; ++b) {
    // prefetch the next block's rows / diagonal (independent of any decision)
    if (b + 1 < nblk) {
#pragma unroll
      for (int q = 0; q < SCAN_ROWS; ++q) {
        int r = (b + 1) * 64 + w * SCAN_ROWS + q;
        nrows[q] = (r < m) ? mask[(size_t)r * NMS_WORDS + lane] : 0ull;
      }
      if (w == 0) { int r = (b + 1) * 64 + lane; nd = (r < m) ? diag[r] : 0ull; }
    }
    if (w == 0) {
      // serial resolve of this block on the scalar unit
      unsigned long long alive = ~rfl64(removed[b]);
      int cnt = __builtin_amdgcn_readfirstlane(s_nkept);
      const int cnt0 = cnt;
      unsigned long long kept = 0;
      uint32_t dlo = (uint32_t)d, dhi = (uint32_t)(d >> 32);
      while (alive != 0 && cnt < K) {
        int i = __builtin_ctzll(alive);
        kept |= 1ull << i;
        ++cnt;
        uint32_t rlo = __builtin_amdgcn_readlane(dlo, i);
        uint32_t rhi = __builtin_amdgcn_readlane(dhi, i);
        alive &= ~(((unsigned long long)rhi << 32) | rlo);
        alive &= ~(1ull << i);
      }
      if ((kept >> lane) & 1ull) {
        unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
        int p = cnt0 + (int)__popcll(kept & lt);
        int c = b * 64 + lane;
        uint32_t oi = sorted_idx[pos0 + c];
        out_idx[p] = (int32_t)oi;
        if (out_boxes) out_boxes[p] = boxes[oi];
        kept_boxes[p] = sboxes[c];
      }
      if (lane == 0) { s_kept64 = kept; s_nkept = cnt; }
    }
    __syncthreads();
    const unsigned long long kept = rfl64(s_kept64);
    const int nk = __builtin_amdgcn_readfirstlane(s_nkept);
    unsigned long long acc = 0;
#pragma unroll
    for (int q = 0; q < SCAN_ROWS; ++q)
      if ((kept >> (w * SCAN_ROWS + q)) & 1ull) acc |= rows[q];
    if (lane > b && acc) atomicOr(&removed[lane], acc);
#pragma unroll
    for (int q = 0; q < SCAN_ROWS; ++q) rows[q] = nrows[q];
    d = nd;
    __syncthreads();
    if (nk >= K) break;
  }
  if (threadIdx.x == 0) {
    int nk = s_nkept;
    int np = pos0 + m;
    st->kept = nk;
    st->pos = np;
    st->done = (nk >= K || np >= st->n_valid) ? 1 : 0;
    *out_count = nk;
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
  char dummy;
  return nms_carve(n, max_output, nullptr, 0, nullptr) + sizeof(dummy) * 0;
}

__global__ void k_nms_export_done(const NmsState* st, int32_t* out_done) { *out_done = st->done; }

// blind_chunks >= 1 chunks are enqueued without looking at the device state (a chunk whose
// predecessor already finished exits at once).  out_done == nullptr: exact mode -- afterwards the
// host reads the state (one sync per further chunk) until the device reports done.
// out_done != nullptr: sync-free mode -- exactly blind_chunks chunks, *out_done tells the caller
// whether the result is complete (it always is when blind_chunks*4096 >= n).
static int nms_run(const float* boxes, const float* scores, int n, int K, float thr, int32_t* out_idx,
                   float* out_boxes, int32_t* out_count, int blind_chunks, int32_t* out_done, void* ws,
                   size_t ws_bytes, hipStream_t st) {
  NmsWorkspace w;
  size_t need = nms_carve(n, K, nullptr, 0, nullptr);
  if (!ws || ws_bytes < need)
    return odet_set_error(ODET_E_WORKSPACE, "odet_nms: workspace too small (%zu < %zu)", ws_bytes, need);
  nms_carve(n, K, ws, ws_bytes, &w);
  uint32_t* sorted = nullptr;
  int rc = odet_sort_pairs_desc(scores, n, w.keys_a, w.vals_a, w.keys_b, w.vals_b, w.hist, &w.state->n_valid,
                                &sorted, st);
  if (rc != ODET_OK) return rc;
  hipLaunchKernelGGL(k_nms_init, dim3(1), dim3(1), 0, st, w.state, out_count);
  ODET_LAUNCH_CHECK();
  const int max_chunks = (n + NMS_CHUNK - 1) / NMS_CHUNK;
  const int nb = (std::min(n, NMS_CHUNK) + 63) / 64;
  if (blind_chunks < 1) blind_chunks = 1;
  for (int c = 0; c < max_chunks; ++c) {
    if (c >= blind_chunks) {
      if (out_done) break;
      // exact mode: need the device's verdict to know whether another chunk is required
      NmsState h;
      ODET_HIP(hipMemcpyAsync(&h, w.state, sizeof(h), hipMemcpyDeviceToHost, st));
      ODET_HIP(hipStreamSynchronize(st));
      if (h.done) break;
    }
    hipLaunchKernelGGL(k_nms_gather, dim3((std::min(n, NMS_CHUNK) + 255) / 256), dim3(256), 0, st, w.state,
                       (const float4*)boxes, sorted, w.sboxes);
    ODET_LAUNCH_CHECK();
    if (c > 0) {
      hipLaunchKernelGGL(k_nms_cross, dim3(NMS_CHUNK / 256), dim3(256), 0, st, w.state, w.sboxes, w.kept_boxes, thr,
                         w.removed_init);
      ODET_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_nms_mask, dim3(nb, nb), dim3(64), 0, st, w.state, w.sboxes, thr, w.mask, w.diag);
    ODET_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_nms_scan, dim3(1), dim3(SCAN_THREADS), 0, st, w.state, w.mask, w.diag, w.removed_init,
                       c > 0 ? 1 : 0, w.sboxes, sorted, (const float4*)boxes, K, out_idx, (float4*)out_boxes,
                       w.kept_boxes, out_count);
    ODET_LAUNCH_CHECK();
  }
  if (out_done) {
    hipLaunchKernelGGL(k_nms_export_done, dim3(1), dim3(1), 0, st, w.state, out_done);
    ODET_LAUNCH_CHECK();
  }
  return ODET_OK;
}

extern "C" int odet_nms(const float* boxes, const float* scores, int n, int max_output, float iou_threshold,
                        int32_t* out_idx, float* out_boxes, int32_t* out_count, int blind_chunks,
                        int32_t* out_done, void* workspace, size_t workspace_bytes, odet_stream_t stream) {
  ODET_REQUIRE(n >= 0 && max_output >= 0, "odet_nms: negative size");
  ODET_REQUIRE(out_count, "odet_nms: null out_count");
  if (n == 0 || max_output == 0) {
    ODET_HIP(hipMemsetAsync(out_count, 0, sizeof(int32_t), (hipStream_t)stream));
    if (out_done) ODET_HIP(hipMemsetAsync(out_done, 1, 1, (hipStream_t)stream));
    return ODET_OK;
  }
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
  if (n == 0 || max_output == 0) {
    ODET_HIP(hipMemsetAsync(out_count, 0, sizeof(int32_t), (hipStream_t)stream));
    if (out_done) ODET_HIP(hipMemsetAsync(out_done, 1, 1, (hipStream_t)stream));
    return ODET_OK;
  }
  ODET_REQUIRE(deltas && anchors && scores && means && stds && out_rois, "odet_region_proposal: null pointer");
  size_t need = odet_region_proposal_workspace_bytes(n, max_output);
  if (!workspace || workspace_bytes < need)
    return odet_set_error(ODET_E_WORKSPACE, "odet_region_proposal: workspace too small (%zu < %zu)",
                          workspace_bytes, need);
  OdetArena ar{(char*)workspace, workspace_bytes, 0};
  float* boxes = ar.take<float>((size_t)n * 4);
  size_t off = odet_align_up(ar.off, 256);
  // region_proposal.py:59 decode + :63 clip (min_edge=None), fused
  int rc = odet_decode(anchors, deltas, 4, n, means, stds, image_h, image_w, boxes, stream);
  if (rc != ODET_OK) return rc;
  // region_proposal.py:73-76 NMS over all n, :81 gather
  int32_t* idx = out_idx ? out_idx : nullptr;
  int32_t* idx_buf = idx;
  if (!idx_buf) {
    idx_buf = reinterpret_cast<int32_t*>((char*)workspace + off);
    off = odet_align_up(off + sizeof(int32_t) * (size_t)max_output, 256);
  }
  return nms_run(boxes, scores, n, max_output, iou_threshold, idx_buf, out_rois, out_count, blind_chunks, out_done,
                 (char*)workspace + off, workspace_bytes - off, (hipStream_t)stream);
}

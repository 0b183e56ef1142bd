// Detection post-processing: model/prediction.py:103-163 post_ops_prediction.
//
// The reference runs a Python loop over the 20 (VOC) / 80 (COCO) foreground classes; each
// iteration dispatches ~12 TF ops, one single-threaded NMS and one device->host sync
// (prediction.py:147).  Here the whole function is ONE launch, a workgroup per class:
//
//   k_postops (grid = num_classes-1 x images, 1024 threads):
//     1. score filter (strict >, :136), decode (:138-140), clip + min-edge filter (:141-143): one
//        RoI per thread, box kept in LDS, 64-bit sort key (score desc, RoI index asc)
//     2. bitonic sort of the keys: one key per thread, compare-exchange distances < 64 by wave
//        shuffles, only the 10 stages with distance >= 64 go through LDS (more than 1024 RoIs:
//        plain LDS bitonic)
//     3. exact greedy NMS (:146) in rounds of 128 sorted candidates: all 16 waves build the
//        128 x 128 lower-triangular suppression bit matrix (and test the round against the boxes
//        kept by earlier rounds), then wave 0 resolves the two 64-blocks with ctz over the alive
//        ballot; stops at max_per_class or when the candidates are exhausted.
//     4. the class's list is published (release fence, then a ticket on a wrapping counter in the
//        workspace); the workgroup that draws the LAST ticket of its image merges: concatenation in
//        class order (:156-158), top-k by (score desc, position asc) (:160), gather (:162), and
//        optionally the fixed-size detection record of the image-parallel all-gather.  The class
//        lists are sorted already, so the merge RANKS instead of sorting again: the rank of an entry
//        is the number of entries before it in (score desc, position asc) order = one binary search
//        per class list.  (Until round 4 the merge was a second launch with a second 1024-key
//        sort: 20.4 + 11.6 us per image in the latency arrangement.)
//
// Output order is the sorted order, one valid instance of tf.nn.top_k(sorted=False)'s
// unspecified order.
#include <mutex>

#include "odet_internal.h"

#define PO_THREADS 1024
#define PO_ROUND 128          // candidates per NMS round

typedef unsigned long long u64;

static int next_pow2_(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}

struct PostOpsParams {    // pointer tables: one entry per image of the batch (blockIdx.y)
  PerImg<const float*> scores_t;     // [R, Ccls]
  PerImg<const float*> deltas_t;     // [R, Ccls, 4]
  PerImg<const float4*> rois_t;      // [R]
  PerImg<const int32_t*> count_dev_t;
  int R, Ccls, P2, K;
  float means[4], stds[4];
  float wmax, hmax, min_edge, score_thr, nms_thr;
  float roi_div;         // rois are divided by this first (im_detect's rois / img_scale); 1 = as they are
  // per-class results (workspace)
  PerImg<int32_t*> cls_count_t;    // [ncls-1]
  PerImg<float4*> cls_boxes_t;     // [ncls-1, K]
  PerImg<float*> cls_scores_t;     // [ncls-1, K]
  PerImg<uint32_t*> ticket_t;      // [1]: class workgroups done (wraps to 0 with the last one)
  // merge
  int ncls1, max_per_image;
  int mode;              // 0: prediction.py top-k (score order); 1: eval loop score-threshold cap (class order)
  PerImg<float4*> out_boxes_t;
  PerImg<int32_t*> out_labels_t;
  PerImg<float*> out_scores_t;
  PerImg<int32_t*> out_count_t;
  PerImg<float*> out_record_t;     // nullable: [max_per_image*6 + 1]
};

__device__ __forceinline__ float key_to_score(uint32_t k) {   // inverse of ~d_float_asc_key
  const uint32_t asc = ~k;
  const uint32_t u = (asc & 0x80000000u) ? (asc & 0x7FFFFFFFu) : ~asc;
  return __uint_as_float(u);
}

struct MergeView {        // one image's pointers
  const int32_t* cls_count; const float4* cls_boxes; const float* cls_scores;
  int ncls1, K, max_per_image, mode;
  float4* out_boxes; int32_t* out_labels; float* out_scores; int32_t* out_count; float* out_record;
};

// The merge of the class lists of one image by ONE workgroup (the last class workgroup to finish): concatenation in class
// order, the max_per_image best by (score desc, position asc).  No second sort: the M-th best order key is found by a
// radix SELECT over the <= 8192 keys (four 8-bit passes on an LDS histogram), ties at that key are taken in position
// order (a block scan: threads hold consecutive slots), and only the <= M selected entries are ranked, among themselves.
// (Measured on the way, latency arrangement: ranking all entries by one binary search per class list 13.5 us -- a chain of
// dependent LDS reads --, the second 1024-key bitonic sort of the old merge launch 11.6 us for the whole launch.)
// LDS: key [nslots] | hist [256] | sel_key [max_per_image] | sel_slot [max_per_image].
__device__ void postops_merge(const MergeView& p, unsigned char* smem) {
  uint32_t* key = reinterpret_cast<uint32_t*>(smem);            // ~asc key of an entry's score: smaller = better
  const int nslots = p.ncls1 * p.K;
  uint32_t* hist = key + nslots;
  uint32_t* sel_key = hist + 256;
  uint32_t* sel_slot = sel_key + (p.max_per_image > 0 ? p.max_per_image : 1);
  __shared__ uint32_t s_prefix, s_target, s_total;
  __shared__ int lds17[17];
  const int tid = threadIdx.x, lane = tid & 63;
  const int per = (nslots + PO_THREADS - 1) / PO_THREADS;       // consecutive slots per thread (slot order = position order)
  const int lo = tid * per;
  // every entry's key and validity in ONE round trip (the class's count is read beside the score, not before it)
  int nvalid = 0;
  for (int q = 0; q < per; ++q) {
    const int slot = lo + q;
    if (slot < nslots) {
      const int c = slot / p.K, k = slot - c * p.K;
      const float sc = p.cls_scores[slot];
      const bool valid = k < min(p.cls_count[c], p.K);
      key[slot] = valid ? ~d_float_asc_key(sc) : 0xFFFFFFFFu;
      nvalid += valid ? 1 : 0;
    }
  }
  int total;
  (void)block_excl_scan(nvalid, lds17, &total);                 // (barriers inside: key[] is complete afterwards)
  // NOTE: an invalid entry carries key 0xFFFFFFFF, which no valid entry can have (it would be a score of -NaN): validity
  // below is `key != 0xFFFFFFFF`
  const int M = (p.mode == 1) ? ((p.max_per_image > 0 && total > p.max_per_image) ? p.max_per_image : total)
                              : min(total, p.max_per_image);                                // prediction.py:160
  // ---- the M-th best key T (1 <= M <= total), by four 8-bit radix passes from the top
  uint32_t T = 0xFFFFFFFFu;
  if (M > 0 && M < total) {
    if (tid == 0) { s_prefix = 0u; s_target = (uint32_t)M; }
    for (int pass = 0; pass < 4; ++pass) {
      const int shift = 24 - 8 * pass;
      if (tid < 256) hist[tid] = 0u;
      __syncthreads();
      const uint32_t prefix = s_prefix;
      const uint32_t himask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
      for (int q = 0; q < per; ++q) {
        const int slot = lo + q;
        if (slot < nslots) {
          const uint32_t k = key[slot];
          if (k != 0xFFFFFFFFu && (k & himask) == prefix) atomicAdd(&hist[(k >> shift) & 255u], 1u);
        }
      }
      __syncthreads();
      if (tid < 64) {
        const uint32_t h0 = hist[4 * tid], h1 = hist[4 * tid + 1], h2 = hist[4 * tid + 2], h3 = hist[4 * tid + 3];
        const int sum = (int)(h0 + h1 + h2 + h3);
        const int inc = wave_incl_scan(sum);
        const int tgt = (int)s_target;
        const bool hit = sum > 0 && inc - sum < tgt && tgt <= inc;
        const u64 bal = __ballot(hit);
        if (bal && lane == __builtin_ctzll(bal)) {
          int run = inc - sum;
          uint32_t d = 4 * tid;
          const uint32_t hh[4] = {h0, h1, h2, h3};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (run < tgt && tgt <= run + (int)hh[j]) { d = 4 * tid + j; s_target = (uint32_t)(tgt - run); }
            run += (int)hh[j];
          }
          s_prefix = prefix | (d << shift);
        }
      }
      __syncthreads();
    }
    T = s_prefix;           // s_target = how many of the entries with key == T belong to the best M (in position order)
  }
  const uint32_t take_ties = (M > 0 && M < total) ? s_target : 0xFFFFFFFFu;
  __syncthreads();
  if (p.mode == 1) {
    // evaluation/pascal_eval_files_utils.py:99-106: when more than max_per_image detections survive,
    // keep those with score >= the max_per_image-th best score (ties at the threshold ALL stay);
    // order = class ascending, NMS order inside a class (the all_boxes[j][i] lists, concatenated).
    int c = 0;
    for (int q = 0; q < per; ++q) {
      const int slot = lo + q;
      if (slot < nslots && key[slot] != 0xFFFFFFFFu && key[slot] <= T) ++c;
    }
    int kept_total;
    int o = block_excl_scan(c, lds17, &kept_total);
    for (int q = 0; q < per; ++q) {
      const int slot = lo + q;
      if (slot < nslots && key[slot] != 0xFFFFFFFFu && key[slot] <= T) {
        p.out_boxes[o] = p.cls_boxes[slot];
        p.out_scores[o] = p.cls_scores[slot];
        p.out_labels[o] = (int32_t)(slot / p.K) + 1;
        ++o;
      }
    }
    if (tid == 0) *p.out_count = kept_total;
    return;
  }
  // ---- mode 0: the selected entries (key < T, and the first take_ties entries with key == T) into a list ...
  int nlt = 0, neq = 0;
  for (int q = 0; q < per; ++q) {
    const int slot = lo + q;
    if (slot < nslots) {
      const uint32_t k = key[slot];
      if (k != 0xFFFFFFFFu) { nlt += (k < T || M == total) ? 1 : 0; neq += (k == T && M < total) ? 1 : 0; }
    }
  }
  int tot_eq, tot_lt;
  int eq_before = block_excl_scan(neq, lds17, &tot_eq);
  int sel_before = block_excl_scan(nlt + max(0, min(neq, (int)min((u64)take_ties, (u64)0x7FFFFFFF) - eq_before)), lds17, &tot_lt);
  {
    int o = sel_before, e = eq_before;
    for (int q = 0; q < per; ++q) {
      const int slot = lo + q;
      if (slot < nslots) {
        const uint32_t k = key[slot];
        bool sel = false;
        if (k != 0xFFFFFFFFu) {
          if (k < T || M == total) sel = true;
          else if (k == T) { sel = (uint32_t)e < take_ties; ++e; }
        }
        if (sel && o < M) { sel_key[o] = k; sel_slot[o] = (uint32_t)slot; ++o; }
      }
    }
  }
  __syncthreads();
  // ... ranked among themselves by (key, slot), written in rank order
  for (int i = tid; i < M; i += PO_THREADS) {
    const uint32_t ki = sel_key[i], si = sel_slot[i];
    int rank = 0;
    for (int j = 0; j < M; ++j) {
      const uint32_t kj = sel_key[j], sj = sel_slot[j];
      rank += (kj < ki || (kj == ki && sj < si)) ? 1 : 0;
    }
    const float4 b = p.cls_boxes[si];
    const float sc = p.cls_scores[si];
    const int lab = (int)(si / (uint32_t)p.K) + 1;
    p.out_boxes[rank] = b;
    p.out_scores[rank] = sc;
    p.out_labels[rank] = lab;
    if (p.out_record) {
      // fixed-size record of the image-parallel all-gather: (x1,y1,x2,y2,score,label), pad score -1
      float* r = p.out_record + (size_t)rank * 6;
      r[0] = b.x; r[1] = b.y; r[2] = b.z; r[3] = b.w; r[4] = sc; r[5] = (float)lab;
    }
  }
  if (p.out_record) {
    for (int i = M + tid; i < p.max_per_image; i += PO_THREADS) {
      float* r = p.out_record + (size_t)i * 6;
      r[0] = 0.0f; r[1] = 0.0f; r[2] = 0.0f; r[3] = 0.0f; r[4] = -1.0f; r[5] = 0.0f;
    }
  }
  if (tid == 0) {
    *p.out_count = M;
    if (p.out_record) p.out_record[(size_t)p.max_per_image * 6] = (float)M;
  }
}

__global__ void __launch_bounds__(PO_THREADS) k_postops(PostOpsParams p) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int img = blockIdx.y;
  const float* __restrict__ in_scores = p.scores_t.v[img];
  const float* __restrict__ in_deltas = p.deltas_t.v[img];
  const float4* __restrict__ in_rois = p.rois_t.v[img];
  const int32_t* __restrict__ in_count = p.count_dev_t.v[img];
  int32_t* __restrict__ cls_count = p.cls_count_t.v[img];
  float4* __restrict__ cls_boxes = p.cls_boxes_t.v[img];
  float* __restrict__ cls_scores = p.cls_scores_t.v[img];
  // layout: keys [max(P2, 2048)] u64 | lbox [R] float4 | sbox [PO_ROUND] float4 | sarea [PO_ROUND] |
  //         kbox [K] float4 | karea [K] float | mask [PO_ROUND][2] u64 | crossf [PO_ROUND] u32
  const int nkeys = p.P2 > 2048 ? p.P2 : 2048;
  u64* keys = reinterpret_cast<u64*>(smem);
  float4* lbox = reinterpret_cast<float4*>(keys + nkeys);
  float4* sbox = lbox + p.R;
  float4* kbox = sbox + PO_ROUND;
  u64* mask = reinterpret_cast<u64*>(kbox + p.K);
  float* sarea = reinterpret_cast<float*>(mask + PO_ROUND * 2);
  float* karea = sarea + PO_ROUND;
  uint32_t* crossf = reinterpret_cast<uint32_t*>(karea + p.K);
  __shared__ int s_nvalid, s_nk, s_last;

  const int c = blockIdx.x + 1;   // class id, prediction.py:135
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (tid == 0) { s_nvalid = 0; s_nk = 0; s_last = 0; }

  // 1. filter + decode + clip.  The row count of the image lives on the device: it is requested TOGETHER with the rows'
  // data (rows up to the static R are always addressable) and only compared afterwards -- one round trip, not two.
  const int Rdev = in_count ? *in_count : p.R;
  u64 mykey = ~0ull;
  for (int r = tid; r < p.P2; r += PO_THREADS) {
    u64 key = ~0ull;
    if (r < p.R) {
      const float s = in_scores[(size_t)r * p.Ccls + c];
      const float4 t = *reinterpret_cast<const float4*>(in_deltas + ((size_t)r * p.Ccls + c) * 4);
      float4 roi = in_rois[r];
      if (r < min(Rdev, p.R) && s > p.score_thr) {                             // :136
        const float d0 = t.x * p.stds[0] + p.means[0];
        const float d1 = t.y * p.stds[1] + p.means[1];
        const float d2 = t.z * p.stds[2] + p.means[2];
        const float d3 = t.w * p.stds[3] + p.means[3];
        if (p.roi_div != 1.0f) {                    // base_fpn_model.py:390 / base_faster_rcnn_model.py:306
          roi.x = roi.x / p.roi_div; roi.y = roi.y / p.roi_div; roi.z = roi.z / p.roi_div; roi.w = roi.w / p.roi_div;
        }
        float4 b = d_decode_box(roi, d0, d1, d2, d3);                          // :138-140
        b = d_clip_box(b, 0.0f, p.wmax, p.hmax);                               // :141-143
        const float e0 = b.z - b.x + 1.0f, e1 = b.w - b.y + 1.0f;              // bbox_tf.py:81-83
        if (e1 >= p.min_edge && e0 >= p.min_edge) {
          lbox[r] = b;
          key = ((u64)(~d_float_asc_key(s)) << 32) | (unsigned)r;
        }
      }
    }
    if (p.P2 > 1024) keys[r] = key; else mykey = key;
  }
  __syncthreads();                                    // (s_nvalid = 0 is visible)
  {
    const u64 bal = __ballot(p.P2 <= 1024 && mykey != ~0ull);
    if (lane == 0 && bal) atomicAdd(&s_nvalid, (int)__popcll(bal));
  }
  __syncthreads();

  // 2. sort: score desc, RoI index asc; rejected rows (key = ~0) go last
  if (p.P2 <= 1024) {
    mykey = bitonic_sort_1024_reg(mykey, keys);
    __syncthreads();
    keys[tid] = mykey;
    __syncthreads();
  } else {
    bitonic_sort_u64(keys, p.P2, PO_THREADS);
    if (tid == 0) {
      // count valid keys by bisection (sorted, ~0 last)
      int lo = 0, hi = p.P2;
      while (lo < hi) { int mid = (lo + hi) >> 1; if (keys[mid] != ~0ull) lo = mid + 1; else hi = mid; }
      s_nvalid = lo;
    }
    __syncthreads();
  }
  const int nvalid = s_nvalid;
  const int K = p.K;

  // 3. greedy NMS in rounds of PO_ROUND sorted candidates
  const u64 lt_lane = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  for (int t0 = 0; t0 < nvalid; t0 += PO_ROUND) {
    const int nk0 = s_nk;                       // kept so far (uniform: read after a barrier)
    if (nk0 >= K) break;
    const int mround = min(PO_ROUND, nvalid - t0);
    // boxes of the round, corner-normalised
    if (tid < PO_ROUND) {
      float4 nb = make_float4(0, 0, 0, 0);
      if (tid < mround) nb = d_norm_box(lbox[(int)(keys[t0 + tid] & 0xFFFFFFFFull)]);
      sbox[tid] = nb;
      sarea[tid] = d_box_area(nb);
      crossf[tid] = 0;
    }
    __syncthreads();
    {
      // thread = (row, slice): 16 columns of the round's lower triangle + 1/8 of the kept list
      const int row = tid & (PO_ROUND - 1), cs = tid >> 7;
      const float4 rb = sbox[row];
      const float ra = sarea[row];
      uint32_t bits = 0;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int col = cs * 16 + j;
        const bool s = (col < row) && d_iou_gt(rb, ra, sbox[col], sarea[col], p.nms_thr);
        bits |= s ? (1u << j) : 0u;
      }
      // 16-bit pieces of the row's 128-bit mask: 4 pieces per u64 word
      reinterpret_cast<unsigned short*>(mask)[row * 8 + cs] = (unsigned short)bits;
      bool sup = false;
      for (int k = cs; k < nk0; k += 8) sup = sup || d_iou_gt(rb, ra, kbox[k], karea[k], p.nms_thr);
      if (sup) atomicOr(&crossf[row], 1u);
    }
    __syncthreads();
    if (w == 0) {
      int nk = nk0;
      u64 kept0 = 0;
#pragma unroll
      for (int blk = 0; blk < 2; ++blk) {
        const int row = blk * 64 + lane;
        const u64 m0 = mask[row * 2 + 0], m1 = mask[row * 2 + 1];
        bool dead = (row >= mround) || (crossf[row] != 0);
        if (blk == 1) dead = dead || ((m0 & kept0) != 0ull);
        const u64 mm = (blk == 0) ? m0 : m1;      // same-block earlier candidates that suppress me
        u64 alive = __ballot(!dead);
        u64 kept = 0;
        while (alive != 0 && nk < K) {
          const int i = __builtin_ctzll(alive);
          kept |= 1ull << i;
          ++nk;
          alive &= ~(1ull << i);
          alive &= ~__ballot((mm >> i) & 1ull);
        }
        if (blk == 0) kept0 = kept;
        if ((kept >> lane) & 1ull) {
          const int slot = nk0 + (blk == 1 ? (int)__popcll(kept0) : 0) + (int)__popcll(kept & lt_lane);
          const u64 kk = keys[t0 + row];
          const float4 ob = lbox[(int)(kk & 0xFFFFFFFFull)];
          kbox[slot] = sbox[row];
          karea[slot] = sarea[row];
          cls_boxes[(size_t)blockIdx.x * K + slot] = ob;
          cls_scores[(size_t)blockIdx.x * K + slot] = key_to_score((uint32_t)(kk >> 32));
        }
      }
      if (lane == 0) s_nk = nk;
    }
    __syncthreads();
  }
  if (tid == 0) cls_count[blockIdx.x] = s_nk;

  // 4. publish this class's list, draw a ticket; the last workgroup of the image merges.  (MI355X guide, "Workgroup
  // dispatch, XCD placement & inter-workgroup visibility": every storing wave drains its stores, workgroup barrier, ONE
  // lane releases at agent scope -- the XCD's L2 writes its dirty lines back -- and only then signals; the consumer
  // acquires at agent scope before any of its loads of the other workgroups' lists.)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // atomicInc wraps to 0 at ncls1 - 1: the counter is clean again for the next call on this workspace
    const uint32_t t = atomicInc(p.ticket_t.v[img], (uint32_t)(p.ncls1 - 1));
    s_last = (t == (uint32_t)(p.ncls1 - 1)) ? 1 : 0;
  }
  __syncthreads();
  if (!s_last) return;
  if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  MergeView mv;
  mv.cls_count = cls_count; mv.cls_boxes = cls_boxes; mv.cls_scores = cls_scores;
  mv.ncls1 = p.ncls1; mv.K = p.K; mv.max_per_image = p.max_per_image; mv.mode = p.mode;
  mv.out_boxes = p.out_boxes_t.v[img]; mv.out_labels = p.out_labels_t.v[img]; mv.out_scores = p.out_scores_t.v[img];
  mv.out_count = p.out_count_t.v[img]; mv.out_record = p.out_record_t.v[img];
  postops_merge(mv, smem);
}

extern "C" size_t odet_post_ops_workspace_bytes(int num_classes, int max_per_class) {
  size_t n1 = (size_t)(num_classes > 1 ? num_classes - 1 : 1);
  size_t k = (size_t)(max_per_class > 0 ? max_per_class : 1);
  return odet_align_up(n1 * 4, 256) + odet_align_up(n1 * k * 16, 256) + odet_align_up(n1 * k * 4, 256) + 1024;
}

// B images in the same launch (blockIdx.y = image); shapes and parameters are common.  ws_clean: the caller promises
// that every image's workspace was zero-filled once and has only ever been used by this function since (the ticket
// counter at its end wraps back to 0 with every call); otherwise the counters are zeroed on the stream first.
int odet_post_ops_batch(const PostOpsImageIO* io, int B, int R, int Ccls, int num_classes, PostOpsExtra ex,
                        const float* means, const float* stds, int max_per_class, int max_per_image,
                        float nms_iou_threshold, float score_threshold, float min_edge, hipStream_t st, int ws_clean) {
  ODET_REQUIRE(io && B >= 1 && B <= ODET_MAX_BATCH, "odet_post_ops: bad batch");
  ODET_REQUIRE(R >= 0 && Ccls > 0 && num_classes >= 1 && num_classes <= Ccls, "odet_post_ops: bad sizes");
  ODET_REQUIRE(max_per_class >= 0 && max_per_image >= 0, "odet_post_ops: negative cap");
  for (int i = 0; i < B; ++i) ODET_REQUIRE(io[i].out_count, "odet_post_ops: null out_count");
  if (R == 0 || num_classes == 1 || max_per_class == 0 || (max_per_image == 0 && ex.mode == 0)) {
    for (int i = 0; i < B; ++i) {
      ODET_HIP(hipMemsetAsync(io[i].out_count, 0, sizeof(int32_t), st));
      if (io[i].out_record && max_per_image > 0) {
        // empty record: pad rows (score -1) are produced by the pack kernel of an empty result
        int rc = odet_pack_detections(io[i].out_boxes, io[i].out_labels, io[i].out_scores, io[i].out_count, 0,
                                      max_per_image, io[i].out_record, (odet_stream_t)st);
        if (rc != ODET_OK) return rc;
      }
    }
    return ODET_OK;
  }
  ODET_REQUIRE(means && stds, "odet_post_ops: null pointer");
  if (R > ODET_POSTOPS_MAX_ROIS)
    return odet_set_error(ODET_E_LIMIT, "odet_post_ops: R %d exceeds %d", R, ODET_POSTOPS_MAX_ROIS);
  const int ncls1 = num_classes - 1;
  if ((int64_t)ncls1 * max_per_class > ODET_POSTOPS_MAX_CANDIDATES)
    return odet_set_error(ODET_E_LIMIT, "odet_post_ops: (num_classes-1)*max_per_class %lld exceeds %d",
                          (long long)ncls1 * max_per_class, ODET_POSTOPS_MAX_CANDIDATES);
  const size_t need = odet_post_ops_workspace_bytes(num_classes, max_per_class);
  PostOpsParams p;
  for (int i = 0; i < ODET_MAX_BATCH; ++i) {
    const PostOpsImageIO& a = io[i < B ? i : 0];
    ODET_REQUIRE(a.scores && a.deltas && a.rois && a.out_boxes && a.out_labels && a.out_scores,
                 "odet_post_ops: null pointer");
    ODET_REQUIRE((uintptr_t)a.deltas % 16 == 0, "odet_post_ops: deltas must be 16-byte aligned");
    if (!a.workspace || a.workspace_bytes < need)
      return odet_set_error(ODET_E_WORKSPACE, "odet_post_ops: workspace too small (%zu < %zu)", a.workspace_bytes, need);
    OdetArena ar{(char*)a.workspace, a.workspace_bytes, 0};
    int32_t* cc = ar.take<int32_t>(ncls1);
    float4* cb = ar.take<float4>((size_t)ncls1 * max_per_class);
    float* cs = ar.take<float>((size_t)ncls1 * max_per_class);
    uint32_t* tk = ar.take<uint32_t>(1);
    ODET_REQUIRE(cc && cb && cs && tk, "odet_post_ops: workspace arena exhausted");
    p.scores_t.v[i] = a.scores; p.deltas_t.v[i] = a.deltas; p.rois_t.v[i] = (const float4*)a.rois;
    p.count_dev_t.v[i] = a.count_dev;
    p.cls_count_t.v[i] = cc; p.cls_boxes_t.v[i] = cb; p.cls_scores_t.v[i] = cs; p.ticket_t.v[i] = tk;
    p.out_boxes_t.v[i] = (float4*)a.out_boxes; p.out_labels_t.v[i] = a.out_labels; p.out_scores_t.v[i] = a.out_scores;
    p.out_count_t.v[i] = a.out_count; p.out_record_t.v[i] = a.out_record;
    if (!ws_clean && i < B) ODET_HIP(hipMemsetAsync(tk, 0, sizeof(uint32_t), st));
  }
  p.R = R; p.Ccls = Ccls; p.P2 = next_pow2_(R < 2 ? 2 : R); p.K = max_per_class;
  for (int k = 0; k < 4; ++k) { p.means[k] = means[k]; p.stds[k] = stds[k]; }
  p.wmax = ex.wmax; p.hmax = ex.hmax; p.roi_div = ex.roi_div;
  p.min_edge = min_edge; p.score_thr = score_threshold; p.nms_thr = nms_iou_threshold;
  p.ncls1 = ncls1; p.max_per_image = max_per_image; p.mode = ex.mode;
  const size_t nkeys1 = (size_t)(p.P2 > 2048 ? p.P2 : 2048);
  const size_t lds_class = nkeys1 * 8 + (size_t)R * 16 + PO_ROUND * 16 + (size_t)max_per_class * 16 + PO_ROUND * 16 +
                           PO_ROUND * 4 + (size_t)max_per_class * 4 + PO_ROUND * 4;
  const size_t lds_merge = (size_t)ncls1 * max_per_class * 4 + 256 * 4 + (size_t)(max_per_image > 0 ? max_per_image : 1) * 8 + 64;
  const size_t lds1 = lds_class > lds_merge ? lds_class : lds_merge;
  if (lds1 > 150 * 1024)
    return odet_set_error(ODET_E_LIMIT, "odet_post_ops: R/max_per_class need %zu B of LDS (> 150 KiB)", lds1);
  static OdetPerDeviceOnce once;    // (executor threads may arrive here together)
  ODET_HIP(once.run([] { return hipFuncSetAttribute((const void*)k_postops, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); }));
  hipLaunchKernelGGL(k_postops, dim3(ncls1, B), dim3(PO_THREADS), lds1, st, p);
  ODET_LAUNCH_CHECK();
  return ODET_OK;
}

static int post_ops_impl(const float* scores, const float* deltas, const float* rois, int R,
                         const int32_t* count_dev, int Ccls, int num_classes, PostOpsExtra ex,
                         const float* means, const float* stds, int max_per_class, int max_per_image,
                         float nms_iou_threshold, float score_threshold, float min_edge, float* out_boxes,
                         int32_t* out_labels, float* out_scores, int32_t* out_count, float* out_record,
                         void* workspace, size_t workspace_bytes, hipStream_t st) {
  PostOpsImageIO io{scores, deltas, rois, count_dev, out_boxes, out_labels, out_scores, out_count, out_record,
                    workspace, workspace_bytes};
  ODET_REQUIRE(out_count, "odet_post_ops: null out_count");
  return odet_post_ops_batch(&io, 1, R, Ccls, num_classes, ex, means, stds, max_per_class, max_per_image,
                             nms_iou_threshold, score_threshold, min_edge, st, /*ws_clean*/ 0);
}

extern "C" int odet_post_ops(const float* scores, const float* deltas, const float* rois, int R,
                             const int32_t* count_dev, int Ccls, int num_classes, int image_h, int image_w,
                             const float* means, const float* stds, int max_per_class, int max_per_image,
                             float nms_iou_threshold, float score_threshold, float min_edge, float* out_boxes,
                             int32_t* out_labels, float* out_scores, int32_t* out_count, void* workspace,
                             size_t workspace_bytes, odet_stream_t stream) {
  ODET_REQUIRE(image_h > 0 && image_w > 0, "odet_post_ops: bad image shape");
  return post_ops_impl(scores, deltas, rois, R, count_dev, Ccls, num_classes,
                       PostOpsExtra{(float)(image_w - 1), (float)(image_h - 1), 1.0f, 0}, means, stds,
                       max_per_class, max_per_image, nms_iou_threshold, score_threshold, min_edge, out_boxes,
                       out_labels, out_scores, out_count, nullptr, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int odet_eval_detect(const float* scores, const float* deltas, const float* rois, int R,
                                const int32_t* count_dev, int Ccls, int num_classes, float img_scale, float raw_h,
                                float raw_w, const float* means, const float* stds, int max_per_class,
                                int max_per_image, float nms_iou_threshold, float score_threshold, float min_size,
                                float* out_boxes, int32_t* out_labels, float* out_scores, int32_t* out_count,
                                void* workspace, size_t workspace_bytes, odet_stream_t stream) {
  ODET_REQUIRE(img_scale > 0.0f && raw_h > 0.0f && raw_w > 0.0f, "odet_eval_detect: bad scale / image size");
  return post_ops_impl(scores, deltas, rois, R, count_dev, Ccls, num_classes,
                       PostOpsExtra{raw_w - 1.0f, raw_h - 1.0f, img_scale, 1}, means, stds, max_per_class,
                       max_per_image, nms_iou_threshold, score_threshold, min_size, out_boxes, out_labels,
                       out_scores, out_count, nullptr, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int odet_post_ops_record(const float* scores, const float* deltas, const float* rois, int R,
                                    const int32_t* count_dev, int Ccls, int num_classes, int image_h, int image_w,
                                    const float* means, const float* stds, int max_per_class, int max_per_image,
                                    float nms_iou_threshold, float score_threshold, float min_edge,
                                    float* out_boxes, int32_t* out_labels, float* out_scores, int32_t* out_count,
                                    float* out_record, void* workspace, size_t workspace_bytes,
                                    odet_stream_t stream) {
  ODET_REQUIRE(out_record, "odet_post_ops_record: null out_record");
  ODET_REQUIRE(image_h > 0 && image_w > 0, "odet_post_ops_record: bad image shape");
  return post_ops_impl(scores, deltas, rois, R, count_dev, Ccls, num_classes,
                       PostOpsExtra{(float)(image_w - 1), (float)(image_h - 1), 1.0f, 0}, means, stds,
                       max_per_class, max_per_image, nms_iou_threshold, score_threshold, min_edge, out_boxes,
                       out_labels, out_scores, out_count, out_record, workspace, workspace_bytes,
                       (hipStream_t)stream);
}

// ---------------------------------------------------------------------- detection records --
// Fixed-size record of one image for the image-parallel all-gather: float32 [max_det, 6] rows
// (x1, y1, x2, y2, score, label), padded rows are zero with score = -1, then one float holding
// the count.  One launch, no host sync.
__global__ void __launch_bounds__(256) k_pack_detections(const float4* __restrict__ boxes,
                                                         const int32_t* __restrict__ labels,
                                                         const float* __restrict__ scores,
                                                         const int32_t* __restrict__ count, int cap, int max_det,
                                                         float* __restrict__ rec) {
  const int m = min(min(*count, cap), max_det);
  for (int i = threadIdx.x; i < max_det; i += 256) {
    float* r = rec + (size_t)i * 6;
    if (i < m) {
      float4 b = boxes[i];
      r[0] = b.x; r[1] = b.y; r[2] = b.z; r[3] = b.w; r[4] = scores[i]; r[5] = (float)labels[i];
    } else {
      r[0] = 0.0f; r[1] = 0.0f; r[2] = 0.0f; r[3] = 0.0f; r[4] = -1.0f; r[5] = 0.0f;
    }
  }
  if (threadIdx.x == 0) rec[(size_t)max_det * 6] = (float)m;
}

extern "C" int odet_pack_detections(const float* boxes, const int32_t* labels, const float* scores,
                                    const int32_t* count_dev, int capacity, int max_det, float* out_record,
                                    odet_stream_t stream) {
  ODET_REQUIRE(boxes && labels && scores && count_dev && out_record, "odet_pack_detections: null pointer");
  ODET_REQUIRE(capacity >= 0 && max_det > 0, "odet_pack_detections: bad sizes");
  hipLaunchKernelGGL(k_pack_detections, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float4*)boxes, labels,
                     scores, count_dev, capacity, max_det, out_record);
  ODET_LAUNCH_CHECK();
  return ODET_OK;
}

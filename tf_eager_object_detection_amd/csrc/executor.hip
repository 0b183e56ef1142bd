// Whole-step descriptor (odet_fpn_step_t) and the native executor: host threads that feed HIP
// streams.  Host-only code (compiled by hipcc with the rest of the library).
//
// Why: the detection hot path is ~11 small launches per image; at ~3 us of host time per launch a
// single enqueuing thread caps throughput near 20k images/s while the GPU, with several images in
// flight on different streams, can go further.  One worker thread per stream removes that cap
// (measured: 4 threads -> 3.3x the launch rate of one).
#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "odet_internal.h"

extern "C" size_t odet_fpn_step_sizeof(void) { return sizeof(odet_fpn_step_t); }

extern "C" int odet_fpn_step_enqueue(const odet_fpn_step_t* s, int stages) {
  ODET_REQUIRE(s, "odet_fpn_step_enqueue: null step");
  int rc = ODET_OK;
  const int norm = s->single_level ? ODET_ROI_NORM_STRIDE : ODET_ROI_NORM_IMAGE;
  const int pool_mode = s->single_level ? s->roi_pool_mode : ODET_ROI_POOL_MAX2;
  if (s->single_level)
    ODET_REQUIRE(s->num_levels == 1 && s->num_maps == 1 && s->sorted_rois == s->rois && !s->roi_level,
                 "odet_fpn_step_enqueue: a single-level step has one level / map, sorted_rois aliasing rois, no roi_level");
  if (stages & ODET_STAGE_PROPOSALS) {
    int fh[ODET_MAX_LEVELS], fw[ODET_MAX_LEVELS], st[ODET_MAX_LEVELS];
    for (int l = 0; l < ODET_MAX_LEVELS; ++l) { fh[l] = s->fh[l]; fw[l] = s->fw[l]; st[l] = s->stride[l]; }
    int32_t* order = s->num_proposals <= ODET_FUSED_ORDER_MAX_ROIS ? s->roi_order : nullptr;
    if (s->single_level) {
      const FpnProposalIO one{s->rpn_logits, s->rpn_deltas, s->rois, s->roi_idx, s->roi_count, nullptr, nullptr, nullptr,
                              nullptr, s->nms_done, s->ws_rpn, s->ws_rpn_bytes, order};
      rc = odet_frcnn_proposals_batch(&one, 1, s->wh, s->A, st[0], fh[0], fw[0], s->image_h, s->image_w, s->rpn_means,
                                      s->rpn_stds, s->num_proposals, s->rpn_nms_iou, s->blind_chunks,
                                      (hipStream_t)s->stream, s->nms_first_chunk, s->ws_rpn_clean);
    } else {
      const FpnProposalIO one{s->rpn_logits, s->rpn_deltas, s->rois, s->roi_idx, s->roi_count, s->sorted_rois, s->roi_level,
                              s->roi_perm, s->level_counts, s->nms_done, s->ws_rpn, s->ws_rpn_bytes, order};
      rc = odet_fpn_proposals_batch(&one, 1, s->num_levels, s->A, fh, fw, st, s->wh, s->image_h, s->image_w, s->rpn_means,
                                    s->rpn_stds, s->num_proposals, s->rpn_nms_iou, s->min_level, s->max_level,
                                    s->blind_chunks, (hipStream_t)s->stream, s->nms_first_chunk, s->ws_rpn_clean);
    }
    if (rc != ODET_OK) return rc;
  }
  if (stages & ODET_STAGE_ROI) {
    if (s->roi_order && s->num_proposals > ODET_FUSED_ORDER_MAX_ROIS) {      // (else: written by the proposal stage)
      rc = odet_roi_order(s->sorted_rois, s->roi_level, s->num_proposals, s->roi_count, s->image_h, s->image_w,
                          s->roi_order, s->stream);
      if (rc != ODET_OK) return rc;
    }
    const RoiImageIO one{s->maps, s->sorted_rois, s->roi_level, s->roi_count, s->roi_order, (float*)s->roi_features};
    rc = odet_roi_pool_batch(&one, 1, s->num_maps, s->channels, s->num_proposals, norm, s->image_h,
                             s->image_w, s->pool_size, pool_mode, (hipStream_t)s->stream,
                             RoiEvents{(hipEvent_t)s->roi_start_event, (hipEvent_t)s->roi_stop_event}, s->maps_f16 ? 1 : 0);
    if (rc != ODET_OK) return rc;
  }
  if (stages & ODET_STAGE_DETECT) {
    ODET_REQUIRE(s->record, "odet_fpn_step_enqueue: null record");
    const PostOpsImageIO one{s->cls_scores, s->cls_deltas, s->sorted_rois, s->roi_count, s->det_boxes, s->det_labels,
                             s->det_scores, s->det_count, s->record, s->ws_post, s->ws_post_bytes};
    rc = odet_post_ops_batch(&one, 1, s->num_proposals, s->ccls, s->num_classes,
                             PostOpsExtra{(float)(s->image_w - 1), (float)(s->image_h - 1), 1.0f, 0}, s->roi_means,
                             s->roi_stds, s->max_per_class, s->max_per_image, s->nms_iou, s->score_threshold,
                             s->min_edge, (hipStream_t)s->stream, s->ws_post_clean);
  }
  return rc;
}

// Several images whose steps share every shape and parameter, in the SAME launches (blockIdx.y =
// image): the per-launch costs (host ~3 us, command processor, kernel-boundary cache maintenance) are
// paid once per batch and the single-workgroup stages of the images run side by side.
static bool same_config(const odet_fpn_step_t* a, const odet_fpn_step_t* b) {
  if (a->image_h != b->image_h || a->image_w != b->image_w || a->num_levels != b->num_levels || a->A != b->A) return false;
  for (int l = 0; l < a->num_levels; ++l)
    if (a->fh[l] != b->fh[l] || a->fw[l] != b->fw[l] || a->stride[l] != b->stride[l]) return false;
  for (int i = 0; i < (a->single_level ? a->A * 4 : a->num_levels * a->A * 2); ++i) if (a->wh[i] != b->wh[i]) return false;
  for (int k = 0; k < 4; ++k)
    if (a->rpn_means[k] != b->rpn_means[k] || a->rpn_stds[k] != b->rpn_stds[k] || a->roi_means[k] != b->roi_means[k] ||
        a->roi_stds[k] != b->roi_stds[k]) return false;
  return a->num_proposals == b->num_proposals && a->rpn_nms_iou == b->rpn_nms_iou && a->min_level == b->min_level &&
         a->max_level == b->max_level && a->blind_chunks == b->blind_chunks && a->nms_first_chunk == b->nms_first_chunk && a->num_maps == b->num_maps &&
         a->channels == b->channels && a->pool_size == b->pool_size && (a->maps_f16 != 0) == (b->maps_f16 != 0) && a->ccls == b->ccls &&
         a->num_classes == b->num_classes && a->max_per_class == b->max_per_class &&
         a->max_per_image == b->max_per_image && a->nms_iou == b->nms_iou &&
         a->score_threshold == b->score_threshold && a->min_edge == b->min_edge && a->stream == b->stream &&
         (a->single_level != 0) == (b->single_level != 0) && a->roi_pool_mode == b->roi_pool_mode &&
         (!a->single_level || a->maps[0].stride == b->maps[0].stride);
}

extern "C" int odet_fpn_step_enqueue_batch(const odet_fpn_step_t* const* steps, int count, int stages) {
  ODET_REQUIRE(steps && count >= 1 && count <= ODET_MAX_BATCH, "odet_fpn_step_enqueue_batch: bad batch");
  const odet_fpn_step_t* s = steps[0];
  ODET_REQUIRE(s, "odet_fpn_step_enqueue_batch: null step");
  if (count == 1) return odet_fpn_step_enqueue(s, stages);
  if (s->single_level)
    for (int i = 0; i < count; ++i)
      ODET_REQUIRE(steps[i] && steps[i]->num_levels == 1 && steps[i]->num_maps == 1 && steps[i]->sorted_rois == steps[i]->rois &&
                   !steps[i]->roi_level, "odet_fpn_step_enqueue_batch: a single-level step has one level / map, sorted_rois "
                   "aliasing rois, no roi_level");
  for (int i = 1; i < count; ++i) {
    ODET_REQUIRE(steps[i], "odet_fpn_step_enqueue_batch: null step");
    ODET_REQUIRE(same_config(s, steps[i]), "odet_fpn_step_enqueue_batch: step %d differs in shape / parameters / stream", i);
  }
  hipStream_t st = (hipStream_t)s->stream;
  int rc = ODET_OK;
  bool clean_all = true;                       // every step's workspaces are promised clean (odet_fpn_step_t.ws_rpn_clean)
  for (int i = 0; i < count; ++i) clean_all = clean_all && steps[i]->ws_rpn_clean != 0;
  if (stages & ODET_STAGE_PROPOSALS) {
    int fh[ODET_MAX_LEVELS], fw[ODET_MAX_LEVELS], sd[ODET_MAX_LEVELS];
    for (int l = 0; l < ODET_MAX_LEVELS; ++l) { fh[l] = s->fh[l]; fw[l] = s->fw[l]; sd[l] = s->stride[l]; }
    FpnProposalIO io[ODET_MAX_BATCH];
    bool ordered_all = true;
    for (int i = 0; i < count; ++i) ordered_all = ordered_all && steps[i]->roi_order != nullptr;
    for (int i = 0; i < count; ++i) {
      const odet_fpn_step_t* t = steps[i];
      io[i] = FpnProposalIO{t->rpn_logits, t->rpn_deltas, t->rois, t->roi_idx, t->roi_count,
                            s->single_level ? nullptr : t->sorted_rois, s->single_level ? nullptr : t->roi_level,
                            s->single_level ? nullptr : t->roi_perm, s->single_level ? nullptr : t->level_counts,
                            t->nms_done, t->ws_rpn, t->ws_rpn_bytes,
                            (ordered_all && s->num_proposals <= ODET_FUSED_ORDER_MAX_ROIS) ? t->roi_order : nullptr};
    }
    if (s->single_level)
      rc = odet_frcnn_proposals_batch(io, count, s->wh, s->A, sd[0], fh[0], fw[0], s->image_h, s->image_w, s->rpn_means,
                                      s->rpn_stds, s->num_proposals, s->rpn_nms_iou, s->blind_chunks, st,
                                      s->nms_first_chunk, clean_all ? 1 : 0);
    else
      rc = odet_fpn_proposals_batch(io, count, s->num_levels, s->A, fh, fw, sd, s->wh, s->image_h, s->image_w,
                                    s->rpn_means, s->rpn_stds, s->num_proposals, s->rpn_nms_iou, s->min_level,
                                    s->max_level, s->blind_chunks, st, s->nms_first_chunk, clean_all ? 1 : 0);
    if (rc != ODET_OK) return rc;
  }
  if (stages & ODET_STAGE_ROI) {
    RoiImageIO io[ODET_MAX_BATCH];
    RoiOrderIO oo[ODET_MAX_BATCH];
    bool ordered = true;
    for (int i = 0; i < count; ++i) ordered = ordered && steps[i]->roi_order != nullptr;
    for (int i = 0; i < count; ++i) {
      const odet_fpn_step_t* t = steps[i];
      io[i] = RoiImageIO{t->maps, t->sorted_rois, t->roi_level, t->roi_count, ordered ? t->roi_order : nullptr,
                         (float*)t->roi_features};
      oo[i] = RoiOrderIO{t->sorted_rois, t->roi_level, t->roi_count, t->roi_order};
    }
    if (ordered && s->num_proposals > ODET_FUSED_ORDER_MAX_ROIS) {             // (else: written by the proposal stage)
      rc = odet_roi_order_batch(oo, count, s->num_proposals, s->image_h, s->image_w, st);
      if (rc != ODET_OK) return rc;
    }
    rc = odet_roi_pool_batch(io, count, s->num_maps, s->channels, s->num_proposals,
                             s->single_level ? ODET_ROI_NORM_STRIDE : ODET_ROI_NORM_IMAGE, s->image_h,
                             s->image_w, s->pool_size, s->single_level ? s->roi_pool_mode : ODET_ROI_POOL_MAX2, st,
                             RoiEvents{(hipEvent_t)s->roi_start_event, (hipEvent_t)s->roi_stop_event},
                             s->maps_f16 ? 1 : 0);
    if (rc != ODET_OK) return rc;
  }
  if (stages & ODET_STAGE_DETECT) {
    bool post_clean = true;                    // (odet_fpn_step_t.ws_post_clean: its own promise since version 101)
    for (int i = 0; i < count; ++i) post_clean = post_clean && steps[i]->ws_post_clean != 0;
    PostOpsImageIO io[ODET_MAX_BATCH];
    for (int i = 0; i < count; ++i) {
      const odet_fpn_step_t* t = steps[i];
      io[i] = PostOpsImageIO{t->cls_scores, t->cls_deltas, t->sorted_rois, t->roi_count, t->det_boxes, t->det_labels,
                             t->det_scores, t->det_count, t->record, t->ws_post, t->ws_post_bytes};
    }
    rc = odet_post_ops_batch(io, count, s->num_proposals, s->ccls, s->num_classes,
                             PostOpsExtra{(float)(s->image_w - 1), (float)(s->image_h - 1), 1.0f, 0}, s->roi_means,
                             s->roi_stds, s->max_per_class, s->max_per_image, s->nms_iou, s->score_threshold,
                             s->min_edge, st, post_clean ? 1 : 0);
  }
  return rc;
}

struct Job { const odet_fpn_step_t* steps[ODET_MAX_BATCH]; int count; int stages; };

struct Worker {
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  std::deque<Job> q;
  bool stop = false;
};

struct odet_exec {
  int device = 0;
  std::vector<Worker*> workers;
  std::mutex done_mu;
  std::condition_variable done_cv;
  long pending = 0;
  int first_error = 0;
  std::string error_text, last_text;
};

static void worker_main(odet_exec* ex, Worker* w) {
  (void)hipSetDevice(ex->device);
  for (;;) {
    Job job;
    {
      std::unique_lock<std::mutex> lk(w->mu);
      w->cv.wait(lk, [&] { return w->stop || !w->q.empty(); });
      if (w->q.empty()) return;          // stop requested and drained
      job = w->q.front();
      w->q.pop_front();
    }
    const int rc = odet_fpn_step_enqueue_batch(job.steps, job.count, job.stages);
    {
      std::lock_guard<std::mutex> lk(ex->done_mu);
      if (rc != ODET_OK && ex->first_error == 0) {
        ex->first_error = rc;
        ex->error_text = odet_last_error();   // thread-local text of this worker
      }
      if (--ex->pending == 0) ex->done_cv.notify_all();
    }
  }
}

extern "C" odet_exec_t* odet_exec_create(int num_workers) {
  if (num_workers < 1 || num_workers > 64) {
    odet_set_error(ODET_E_INVALID, "odet_exec_create: num_workers %d out of range", num_workers);
    return nullptr;
  }
  odet_exec* ex = new odet_exec();
  if (hipGetDevice(&ex->device) != hipSuccess) ex->device = 0;
  for (int i = 0; i < num_workers; ++i) {
    Worker* w = new Worker();
    ex->workers.push_back(w);
    w->th = std::thread(worker_main, ex, w);
  }
  return ex;
}

extern "C" void odet_exec_destroy(odet_exec_t* ex) {
  if (!ex) return;
  for (Worker* w : ex->workers) {
    { std::lock_guard<std::mutex> lk(w->mu); w->stop = true; }
    w->cv.notify_all();
  }
  for (Worker* w : ex->workers) {
    if (w->th.joinable()) w->th.join();
    delete w;
  }
  delete ex;
}

extern "C" int odet_exec_submit_batch(odet_exec_t* ex, int worker, const odet_fpn_step_t* const* steps, int count,
                                      int stages) {
  ODET_REQUIRE(ex && steps, "odet_exec_submit: null pointer");
  ODET_REQUIRE(count >= 1 && count <= ODET_MAX_BATCH, "odet_exec_submit: batch %d out of range", count);
  ODET_REQUIRE(worker >= 0 && worker < (int)ex->workers.size(), "odet_exec_submit: worker %d out of range", worker);
  Job job;
  for (int i = 0; i < ODET_MAX_BATCH; ++i) job.steps[i] = (i < count) ? steps[i] : nullptr;
  for (int i = 0; i < count; ++i) ODET_REQUIRE(steps[i], "odet_exec_submit: null step");
  job.count = count;
  job.stages = stages;
  { std::lock_guard<std::mutex> lk(ex->done_mu); ++ex->pending; }
  Worker* w = ex->workers[worker];
  { std::lock_guard<std::mutex> lk(w->mu); w->q.push_back(job); }
  w->cv.notify_one();
  return ODET_OK;
}

extern "C" int odet_exec_submit(odet_exec_t* ex, int worker, const odet_fpn_step_t* step, int stages) {
  return odet_exec_submit_batch(ex, worker, &step, 1, stages);
}

extern "C" int odet_exec_wait(odet_exec_t* ex) {
  ODET_REQUIRE(ex, "odet_exec_wait: null executor");
  std::unique_lock<std::mutex> lk(ex->done_mu);
  ex->done_cv.wait(lk, [&] { return ex->pending == 0; });
  const int rc = ex->first_error;
  ex->last_text = ex->error_text;
  ex->first_error = 0;
  ex->error_text.clear();
  return rc;
}

extern "C" const char* odet_exec_last_error(odet_exec_t* ex) { return ex ? ex->last_text.c_str() : ""; }

/* odet.h -- C ABI of libodet_hip.so: the MI355X (gfx950) detection hot path.
 *
 * Drop-in boundary for the Faster-R-CNN / FPN inference hot path of
 * irvingzhang0512/tf_eager_object_detection.  The reference has no FFI of its own (it is
 * 100 % Python over TensorFlow ops); each entry point below replaces the TensorFlow work
 * behind one reference function, cited as file:line relative to the reference checkout.
 * The Python package tf_eager_object_detection_amd re-creates the reference's call surface
 * on top of these symbols with ctypes (see INTEGRATION.md for the binding).
 *
 * Conventions
 *  - extern "C", plain pointers + sizes, no C++/torch types.  All array pointers are DEVICE
 *    pointers (HBM) unless the parameter is documented "host".  The caller owns every
 *    buffer; the library allocates nothing.  Scratch memory is a caller-provided workspace
 *    sized by the matching *_workspace_bytes() query.
 *  - `stream` is a hipStream_t passed as void* (0 = null stream).  Calls only enqueue work;
 *    the few that must read a device-side count to decide how much more work to enqueue say
 *    so ("host-syncs on `stream`").
 *  - Boxes are float32 [x1,y1,x2,y2] rows; feature maps are NHWC float32 with batch = 1.
 *  - Return value: 0 = ok, negative = error (ODET_E_*); odet_last_error() gives the text for
 *    the calling thread.  Nothing throws across the ABI.
 *  - Arithmetic is float32 in the reference's operation order, no FMA contraction; exp/log
 *    are correctly rounded float32.  Ties in NMS / top-k: (score desc, index asc).
 */
#ifndef ODET_H_
#define ODET_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 100: round 4.  101: odet_fpn_step_t.ws_post_clean, the odet_*_x3 entry points (three bfloat16 limbs) with their split-K
 * workspace.  102: the odet_*_x2 entry points (two float16 limbs) and odet_split_f16x2; odet_bias_relu_maxpool keeps a NaN
 * in float32.  103: the two-limb launches report an out-of-range activation in a status word of their workspace
 * (odet_x2_status_offset); the tile-forcing diagnostics left this header and the shipped library (include/odet_diag.h, a
 * separate -DODET_DIAG build). */
#define ODET_VERSION 103

#define ODET_OK 0
#define ODET_E_INVALID (-1)   /* bad argument (null pointer, negative size, unsupported shape) */
#define ODET_E_WORKSPACE (-2) /* workspace too small */
#define ODET_E_HIP (-3)       /* a HIP runtime call failed */
#define ODET_E_LIMIT (-4)     /* size exceeds a documented kernel limit */

typedef void* odet_stream_t;

int odet_version(void);
const char* odet_last_error(void);

/* ---- anchors ------------------------------------------------------------------------ */

/* utils/anchor_generator.py:46-60 generate_by_anchor_base_tf.
 * out[(y*fw+x)*A+a] = base[a] + (x*stride, y*stride, x*stride, y*stride).
 * anchor_base: device [A,4]; out: device [fh*fw*A,4]. */
int odet_anchors_shift(const float* anchor_base, int A, int feat_stride, int fh, int fw,
                       float* out, odet_stream_t stream);

#define ODET_MAX_LEVELS 8
#define ODET_MAX_ANCHORS_PER_CELL 32
/* utils/anchor_generator.py:137-178 make_anchors for ALL pyramid levels in one launch
 * (caller loop: model/fpn/base_fpn_model.py:163-186 _get_anchors).
 * Host arrays: fh/fw/stride [num_levels]; wh [num_levels*A*2] = per level, per anchor
 * (w, h) in float32 exactly as enum_scales/enum_ratios produce them (the Python layer
 * evaluates that A-element table with IEEE float32 ops).  out: device [sum fh*fw*A, 4],
 * levels concatenated in order, location-major / anchor-minor inside a level. */
int odet_anchors_fpn(int num_levels, int A, const int* fh, const int* fw, const int* stride,
                     const float* wh, float* out, odet_stream_t stream);

/* ---- box transforms ------------------------------------------------------------------ */

/* utils/bbox_transform.py:32-55 decode_bbox_with_mean_and_std.  means/stds: host [4].
 * If clip_h > 0 the result is also clipped to [0, clip_w-1] x [0, clip_h-1] as
 * utils/bbox_tf.py:70-74 does with min_value 0 (fusion used by region_proposal.py:59-63).
 * delta_stride: floats between consecutive delta rows (4 for a dense [n,4] array). */
int odet_decode(const float* anchors, const float* deltas, int64_t delta_stride, int n,
                const float* means, const float* stds, int clip_h, int clip_w, float* out,
                odet_stream_t stream);

/* utils/bbox_transform.py:4-29 encode_bbox_with_mean_and_std. */
int odet_encode(const float* src, const float* dst, int n, const float* means, const float* stds,
                float* out, odet_stream_t stream);

/* utils/bbox_tf.py:59-78 bboxes_clip_filter with min_edge=None: clip only. */
int odet_clip(const float* boxes, int n, float min_value, int max_h, int max_w, float* out,
              odet_stream_t stream);

size_t odet_compact_workspace_bytes(int n);
/* utils/bbox_tf.py:59-84 bboxes_clip_filter with min_edge: clip, keep rows with both
 * (+1) edges >= min_edge, ascending index order.  out_boxes [n,4], out_idx int64 [n],
 * out_count device int32[1]. */
int odet_clip_filter(const float* boxes, int n, float min_value, int max_h, int max_w,
                     float min_edge, float* out_boxes, int64_t* out_idx, int32_t* out_count,
                     void* workspace, size_t workspace_bytes, odet_stream_t stream);

/* utils/bbox_tf.py:87-101 bboxes_range_filter: indices of boxes fully inside the image. */
int odet_range_filter(const float* boxes, int n, int max_h, int max_w, int64_t* out_idx,
                      int32_t* out_count, void* workspace, size_t workspace_bytes,
                      odet_stream_t stream);

/* model/prediction.py:136 tf.where(score > thr): ascending indices of values[i*stride] > thr. */
int odet_where_greater(const float* values, int64_t stride, int n, float thr, int64_t* out_idx,
                       int32_t* out_count, void* workspace, size_t workspace_bytes,
                       odet_stream_t stream);

/* utils/bbox_tf.py:37-56 pairwise_iou (+1 convention, 0 where intersection == 0).
 * out: device [n,m] row-major. */
int odet_pairwise_iou(const float* boxes1, int n, const float* boxes2, int m, float* out,
                      odet_stream_t stream);

/* rows gather: out[i,:] = src[idx[i],:] for i < count (count_dev overrides n when non-null).
 * idx_is_64 selects int64 / int32 indices.  row_floats floats per row. */
int odet_gather_rows(const float* src, const void* idx, int idx_is_64, int n,
                     const int32_t* count_dev, int row_floats, float* out, odet_stream_t stream);

/* ---- RPN score glue ------------------------------------------------------------------ */

#define ODET_RPN_LAYOUT_FPN 0   /* model/fpn/base_fpn_model.py:223,429: [n,2] (bg,fg) pairs */
#define ODET_RPN_LAYOUT_FRCNN 1 /* model/faster_rcnn/base_faster_rcnn_model.py:149-152:
                                   per location [A bg | A fg] */
/* fg probability = softmax(bg,fg)[1] with tf.nn.softmax arithmetic.  logits: [nloc, 2*A]
 * (FRCNN) or [n,2] with A ignored (FPN, nloc = n).  out: [nloc*A] resp. [n]. */
int odet_rpn_fg_softmax(const float* logits, int nloc, int A, int layout, float* out,
                        odet_stream_t stream);

/* ---- NMS / region proposal ----------------------------------------------------------- */

size_t odet_nms_workspace_bytes(int n, int max_output);
/* tf.image.non_max_suppression (NonMaxSuppressionV3, score_threshold=-inf) as called at
 * model/region_proposal.py:74-76 and model/prediction.py:146: exact greedy NMS over all n
 * boxes, IoU without +1, strict '>', stop at max_output.  out_idx int32 [max_output]
 * (original indices in keep order), out_boxes (nullable) [max_output,4] gathered rows,
 * out_count device int32[1].
 * Candidates are consumed in (score desc, index asc) order in chunks: chunk 0 = the best
 * ~1.5*max_output candidates (radix select, no full sort); if it does not reach max_output the
 * remaining candidates are fully sorted and consumed in chunks of 4096.  `blind_chunks` (>= 1)
 * chunks are enqueued unconditionally (the work of a chunk after completion is skipped on the
 * device).
 *  - out_done == NULL  (exact mode): if more chunks may be needed the call host-syncs on
 *    `stream` once per further chunk until the device reports completion.  Always exact.
 *  - out_done != NULL  (sync-free mode, graph-capturable): exactly blind_chunks chunks run;
 *    *out_done (device int32) = 1 when the result is complete, 0 when max_output was not
 *    reached within them; odet_nms / odet_region_proposal then hold the exact PREFIX found so far
 *    (out_count of it), the caller re-runs in exact mode.  The fused stages whose count feeds further
 *    kernels on the device (odet_fpn_proposals, odet_frcnn_proposals, the step descriptor) report an
 *    incomplete result as ZERO proposals (*out_count = 0, level counts 0, *out_done = 0): nothing
 *    downstream ever runs on a partial or stale RoI list -- an image that did not complete yields no
 *    detections and says so. */
int odet_nms(const float* boxes, const float* scores, int n, int max_output, float iou_threshold,
             int32_t* out_idx, float* out_boxes, int32_t* out_count, int blind_chunks,
             int32_t* out_done, void* workspace, size_t workspace_bytes, odet_stream_t stream);

size_t odet_region_proposal_workspace_bytes(int n, int max_output);
/* model/region_proposal.py:55-81 RegionProposal.call: decode (means/stds host [4]) -> clip
 * to the image -> NMS over ALL n anchors -> gather.  out_rois [max_output,4], out_idx
 * (nullable) int32 [max_output], out_count device int32[1]; blind_chunks / out_done as in
 * odet_nms. */
int odet_region_proposal(const float* deltas, const float* anchors, const float* scores, int n,
                         int image_h, int image_w, const float* means, const float* stds,
                         int max_output, float iou_threshold, float* out_rois, int32_t* out_idx,
                         int32_t* out_count, int blind_chunks, int32_t* out_done, void* workspace,
                         size_t workspace_bytes, odet_stream_t stream);

size_t odet_fpn_proposals_workspace_bytes(int n, int max_output);
/* The whole proposal stage of model/fpn/base_fpn_model.py BaseFPN.call in one entry point:
 * :220 _get_anchors (utils/anchor_generator.py:137-178, anchors are produced in registers and
 * never stored), :223 softmax(rpn_score)[:,1], :224 RegionProposal (model/region_proposal.py:
 * 55-81) and, when out_sorted_rois != NULL, :256 _assign_levels (:303-324) fused into the last
 * NMS launch.  rpn_logits [n,2] (bg,fg) as RpnHead emits them (:429), rpn_deltas [n,4], levels
 * concatenated in list order; fh/fw/stride/wh host arrays as in odet_anchors_fpn
 * (n = sum fh*fw*A).  Outputs as odet_region_proposal + odet_assign_levels.  out_order (nullable,
 * int32 [max_output], needs out_sorted_rois and max_output <= ODET_FUSED_ORDER_MAX_ROIS): the spatial
 * processing order of the level-sorted RoIs for odet_roi_pool_ordered (what odet_roi_order computes),
 * produced by the tail of the NMS walk instead of a launch of its own. */
#define ODET_FUSED_ORDER_MAX_ROIS 1024
int odet_fpn_proposals(const float* rpn_logits, const float* rpn_deltas, int num_levels, int A,
                       const int* fh, const int* fw, const int* stride, const float* wh,
                       int image_h, int image_w, const float* means, const float* stds,
                       int max_output, float iou_threshold, int min_level, int max_level,
                       float* out_rois, int32_t* out_idx, int32_t* out_count,
                       float* out_sorted_rois, int32_t* out_level, int64_t* out_perm,
                       int32_t* out_level_counts, int32_t* out_order, int blind_chunks,
                       int32_t* out_done, void* workspace, size_t workspace_bytes, odet_stream_t stream);

size_t odet_frcnn_proposals_workspace_bytes(int n, int max_output);
/* The proposal stage of model/faster_rcnn/base_faster_rcnn_model.py BaseFasterRcnn.call in one entry
 * point: :139-142 generate_by_anchor_base_tf (utils/anchor_generator.py:46-60, anchors produced in
 * registers), :149-152 fg probability from the [A bg | A fg] channel layout, :153 RegionProposal
 * (model/region_proposal.py:55-81).  rpn_logits [fh*fw, 2A], rpn_deltas [fh*fw*A, 4],
 * anchor_base host [A,4] float32 (generate_anchor_base cast to float32, :84).  n = fh*fw*A. */
int odet_frcnn_proposals(const float* rpn_logits, const float* rpn_deltas, const float* anchor_base,
                         int A, int feat_stride, int fh, int fw, int image_h, int image_w,
                         const float* means, const float* stds, int max_output,
                         float iou_threshold, float* out_rois, int32_t* out_idx,
                         int32_t* out_count, int blind_chunks, int32_t* out_done, void* workspace,
                         size_t workspace_bytes, odet_stream_t stream);

/* ---- FPN level assignment ------------------------------------------------------------ */

#define ODET_ASSIGN_MAX_ROIS 8192
/* model/fpn/base_fpn_model.py:303-324 _assign_levels.  rois [n,4] (count_dev overrides n
 * when non-null; n is then the capacity).  Outputs: out_rois [n,4] level-sorted (stable),
 * out_level int32 [n] (level - min_level of each SORTED row), out_perm int64 [n] (original
 * row of each sorted row), out_counts int32 [max_level-min_level+1]. */
int odet_assign_levels(const float* rois, int n, const int32_t* count_dev, int min_level,
                       int max_level, float* out_rois, int32_t* out_level, int64_t* out_perm,
                       int32_t* out_counts, odet_stream_t stream);

/* ---- RoI feature extraction ---------------------------------------------------------- */

#define ODET_ROI_NORM_STRIDE 0   /* model/roi_pooling.py:63-74: (roi/stride)/(dim-1) */
#define ODET_ROI_NORM_IMAGE 1    /* model/roi_pooling.py:25-35: roi/image_size (FPN) */
#define ODET_ROI_NORM_TP_ALIGN 2 /* model/roi_pooling.py:93-137,174-177: tensorpack RoIAlign */
#define ODET_ROI_NORM_TP_ALIGN_NOPAD 3 /* same with pad_border=False (roi_pooling.py:93,97) */
#define ODET_ROI_POOL_NONE 0     /* crop P x P            (roi_pooling.py:85-90) */
#define ODET_ROI_POOL_MAX2 1     /* crop 2P x 2P + 2x2 max (roi_pooling.py:36-42,75-84) */
#define ODET_ROI_POOL_AVG2 2     /* crop 2P x 2P + 2x2 avg (roi_pooling.py:149-154) */

typedef struct {
  const float* data; /* device NHWC [1,H,W,C] */
  int32_t H, W;
  float stride; /* used by NORM_STRIDE / NORM_TP_ALIGN */
} odet_level_t;

/* tf.image.crop_and_resize (bilinear, extrapolation 0) + optional 2x2 pool, fused, over up
 * to ODET_MAX_LEVELS feature maps in one launch.  levels: host array; roi_level: device
 * int32 [n] index into levels for each RoI (nullable = all 0); count_dev (nullable)
 * overrides n.  C % 4 == 0.  out [n,P,P,C]. */
int odet_roi_pool(const odet_level_t* levels, int num_levels, int C, const float* rois,
                  const int32_t* roi_level, int n, const int32_t* count_dev, int norm_mode,
                  int image_h, int image_w, int pool_size, int pool_mode, float* out,
                  odet_stream_t stream);

/* odet_roi_pool with HIP events attached to the dispatch itself (hipExtLaunchKernel start/stop
 * events, created with odet_prof_event_create): the kernel's own begin / end timestamps, for
 * bench.py's roofline figure.  odet_prof_event_elapsed_ms host-syncs on the stop event. */
int odet_roi_pool_timed(const odet_level_t* levels, int num_levels, int C, const float* rois,
                        const int32_t* roi_level, int n, const int32_t* count_dev, int norm_mode,
                        int image_h, int image_w, int pool_size, int pool_mode, float* out,
                        odet_stream_t stream, void* start_event, void* stop_event);
int odet_prof_event_create(void** ev);
/* Measurement infrastructure for bench.py's roofline object (no stage of the reference; csrc/calib.hip): a kernel
 * that only moves bytes -- reads read_bytes of src once (1 KB rows, XCD x reads the x-th eighth) and writes write_bytes
 * of dst with the RoI kernel's store instruction, interleaved at that byte ratio.  Timed with the optional events
 * (nullable), it is the time this box's memory system needs for the RoI launch's read : write mix. */
int odet_calib_stream_mix(const void* src, unsigned long long read_bytes, void* dst, unsigned long long write_bytes,
                          odet_stream_t stream, void* start_event, void* stop_event);
/* Counter calibration (measurement infrastructure): reads (an odd number of) 64 * bytes_per_lane-byte rows of src, `bytes`
 * in all, exactly once with 8 or 16 bytes per lane in a permuted row order -- a known byte count in the access shape of
 * the float16 / float32 RoI kernels, against which FETCH_SIZE's gfx950 factor is measured in the same profiler pass
 * (kernel names k_calib_read<8> / k_calib_read<16>).  Returns the rows read through *sink only in name (never written). */
int odet_calib_read_rows(const void* src, unsigned long long bytes, int bytes_per_lane, void* sink, odet_stream_t stream);
int odet_prof_event_destroy(void* ev);
int odet_prof_event_elapsed_ms(void* start, void* stop, float* ms);

/* Spatial processing order for odet_roi_pool_ordered: out_order int32 [n] = the RoI rows sorted by
 * (level, y centre, x centre) (padded rows >= count last).  Native addition (no reference counterpart):
 * output row r still holds RoI r, only the order in which workgroups pick RoIs changes, so that the
 * chunk of RoIs one XCD processes taps one band of one pyramid level (fewer lines fetched into several
 * L2s, sliding working set).  n <= 8192. */
int odet_roi_order(const float* rois, const int32_t* roi_level, int n, const int32_t* count_dev,
                   int image_h, int image_w, int32_t* out_order, odet_stream_t stream);
/* odet_roi_pool processing the RoIs in `order` (nullable = index order); start/stop events nullable
 * (see odet_roi_pool_timed). */
int odet_roi_pool_ordered(const odet_level_t* levels, int num_levels, int C, const float* rois,
                          const int32_t* roi_level, int n, const int32_t* count_dev,
                          const int32_t* order, int norm_mode, int image_h, int image_w,
                          int pool_size, int pool_mode, float* out, odet_stream_t stream,
                          void* start_event, void* stop_event);

/* odet_roi_pool_ordered for float16 feature maps (BASELINE config 5): levels[i].data points at NHWC
 * float16, out is float16 [n,P,P,C]; boxes stay float32, taps are widened to float32, lerped and pooled in
 * the same operation order, and rounded to nearest-even on the store.  Pooled (max / avg) un-padded modes
 * with pool_size <= 8. */
int odet_roi_pool_f16(const odet_level_t* levels, int num_levels, int C, const float* rois,
                      const int32_t* roi_level, int n, const int32_t* count_dev, const int32_t* order,
                      int norm_mode, int image_h, int image_w, int pool_size, int pool_mode, void* out,
                      odet_stream_t stream);
/* the same launch with HIP events attached to the dispatch (see odet_roi_pool_timed) */
int odet_roi_pool_f16_timed(const odet_level_t* levels, int num_levels, int C, const float* rois,
                            const int32_t* roi_level, int n, const int32_t* count_dev,
                            const int32_t* order, int norm_mode, int image_h, int image_w, int pool_size,
                            int pool_mode, void* out, odet_stream_t stream, void* start_event,
                            void* stop_event);

/* ---- detection post-processing ------------------------------------------------------- */

#define ODET_POSTOPS_MAX_ROIS 4096
#define ODET_POSTOPS_MAX_CANDIDATES 8192 /* (num_classes-1) * max_per_class */
size_t odet_post_ops_workspace_bytes(int num_classes, int max_per_class);
/* model/prediction.py:103-163 post_ops_prediction in ONE launch: a workgroup per class (filter, decode, clip,
 * sort, exact NMS); the last class workgroup of an image to finish merges the lists (concatenation, top-k) -- no
 * second launch, no host sync.  scores [R,Ccls] softmax, deltas [R,Ccls,4] (16-byte aligned), rois [R,4]
 * (count_dev overrides R).
 * Loops classes 1..num_classes-1 (num_classes <= Ccls).  means/stds host [4].
 * Outputs (capacity max_per_image): out_boxes [.,4], out_labels int32, out_scores in
 * (score desc, class asc, NMS order) order, out_count device int32[1]
 * (0 == the reference returns (None, None, None)). */
int odet_post_ops(const float* scores, const float* deltas, const float* rois, int R,
                  const int32_t* count_dev, int Ccls, int num_classes, int image_h, int image_w,
                  const float* means, const float* stds, int max_per_class, int max_per_image,
                  float nms_iou_threshold, float score_threshold, float min_edge,
                  float* out_boxes, int32_t* out_labels, float* out_scores, int32_t* out_count,
                  void* workspace, size_t workspace_bytes, odet_stream_t stream);

/* odet_post_ops that also writes the fixed-size detection record of odet_pack_detections
 * (out_record float32 [max_per_image*6 + 1]) from the same merge launch. */
int odet_post_ops_record(const float* scores, const float* deltas, const float* rois, int R,
                         const int32_t* count_dev, int Ccls, int num_classes, int image_h,
                         int image_w, const float* means, const float* stds, int max_per_class,
                         int max_per_image, float nms_iou_threshold, float score_threshold,
                         float min_edge, float* out_boxes, int32_t* out_labels, float* out_scores,
                         int32_t* out_count, float* out_record, void* workspace,
                         size_t workspace_bytes, odet_stream_t stream);

/* evaluation/pascal_eval_files_utils.py:76-106 (the mAP-producing per-image loop; same loop inlined
 * at scripts/eval_coco.py:117-164) on the outputs of im_detect (model/fpn/base_fpn_model.py:364-390,
 * model/faster_rcnn/base_faster_rcnn_model.py:279-306): rois (resized-image pixels) are divided by
 * img_scale, then per class 1..num_classes-1: score > score_threshold -> decode -> clip to the RAW
 * image (raw_h, raw_w) with the min_size edge filter -> NMS(max_per_class, nms_iou_threshold); then
 * the per-image cap: when more than max_per_image detections survive, those with score >= the
 * max_per_image-th best score stay (ties all stay; max_per_image <= 0 disables the cap).
 * Outputs in the order of the reference's all_boxes lists (class ascending, NMS order inside a
 * class), capacity (num_classes-1)*max_per_class rows; out_count device int32[1].
 * Workspace: odet_post_ops_workspace_bytes. */
int odet_eval_detect(const float* scores, const float* deltas, const float* rois, int R,
                     const int32_t* count_dev, int Ccls, int num_classes, float img_scale,
                     float raw_h, float raw_w, const float* means, const float* stds,
                     int max_per_class, int max_per_image, float nms_iou_threshold,
                     float score_threshold, float min_size, float* out_boxes, int32_t* out_labels,
                     float* out_scores, int32_t* out_count, void* workspace,
                     size_t workspace_bytes, odet_stream_t stream);

/* ---- FPN neck: top-down merge -------------------------------------- */

/* model/fpn/resnet_fpn.py:385-398 (ResnetFpnNeck.call): P_k = Add([resize_bilinear(P_{k+1},
 * size(C_k)) * 0.5, lateral(C_k) * 0.5]) with tf.image.resize_bilinear of TF 1.x
 * (align_corners=False: src = dst * in/out, lo = floor, hi = min(lo + 1, in - 1)) in one
 * launch.  NHWC: top [B,h,w,C], lateral and out [B,H,W,C]; f16 != 0: float16 maps (float32
 * arithmetic, one rounding at the end), C % 8 == 0; otherwise float32, C % 4 == 0. */
int odet_fpn_topdown_merge(const void* top, int h, int w, const void* lateral, int H, int W,
                           int B, int C, void* out, int f16, odet_stream_t stream);

/* Convolution epilogue of the dense path, in place on an NHWC activation x [npix, C]:
 * x = relu?((x + bias[C]) (+ residual[npix, C])) -- the frozen-BatchNormalization bias, the
 * bottleneck's Add([shortcut, x]) and Activation('relu') of model/fpn/resnet_fpn.py:154-205
 * (and the RpnHead's ReLU, base_fpn_model.py:393-434) in one pass.  residual may be NULL.
 * f16 != 0: float16 tensors (float32 arithmetic, one rounding), C % 8 == 0; else float32, C % 4 == 0. */
int odet_bias_act(void* x, const void* bias, const void* residual, long long npix, int C, int relu,
                  int f16, odet_stream_t stream);

/* Convolution epilogue + max-pooling in one pass (model/fpn/resnet_fpn.py:228-259 conv1 -> relu -> pool1;
 * model/faster_rcnn/vgg16_faster_rcnn.py:260-342 conv -> relu -> MaxPooling2D((2,2), 2, 'same')):
 * out[B, OH, OW, C] = maxpool_{kernel, stride, pad}(relu(x[B, H, W, C] + bias[C])), NHWC, x = the convolution
 * WITHOUT its bias.  Window taps outside the map are skipped (= zero padding after a ReLU, = TF 'same').
 * Bit-identical to the separate passes (bias and ReLU commute with the maximum).  f16: C % 8 == 0; else C % 4. */
int odet_bias_relu_maxpool(const void* x, const void* bias, void* out, int B, int H, int W, int C, int OH, int OW,
                           int kernel, int stride, int pad, int f16, odet_stream_t stream);

/* 3x3 stride-1 'same' convolution as an implicit GEMM on the matrix cores -- the RpnHead's convolution
 * (model/fpn/base_fpn_model.py:401-417, model/faster_rcnn/base_faster_rcnn_model.py:315-321): x NHWC float16
 * [batch,H,W,cin], w float16 [cout][3][3][cin] (= a channels_last torch weight), y NHWC float16 [batch,H,W,cout],
 * float32 accumulation; bias (nullable, float16 [cout]) and relu are applied before the one rounding.
 * cin % 64 == 0, cout % 64 == 0 (256-channel workgroup tiles; 128 / 64-channel tiles when cout is not a multiple of 256). */
int odet_conv3x3_f16(const void* x, const void* w, const void* bias, void* y, int batch, int H, int W,
                     int cin, int cout, int relu, odet_stream_t stream);
/* the same convolution (shared weights) over up to ODET_MAX_LEVELS maps in ONE launch -- the RpnHead over the pyramid
 * levels (base_fpn_model.py:188-200): the workgroups of the small levels fill the tail of the big ones'.  levels: host
 * array, x / y NHWC float16 [batch,H,W,cin] / [batch,H,W,cout]; list the big levels first. */
typedef struct {
  const void* x; void* y;
  int32_t H, W;
} odet_conv_level_t;
/* A stage's last convolution of VGG16 with its pooling (vgg16_faster_rcnn.py:260-342: Conv2D(3x3, 'same') + ReLU +
 * MaxPooling2D((2, 2), 2, padding='same')) in one launch: y [batch][ceil(H/2)][ceil(W/2)][cout] float16 -- the un-pooled map
 * is never written.  Same operand rules as odet_conv3x3_f16; max commutes with the rounding, so the result equals pooling
 * the rounded map. */
int odet_conv3x3_relu_pool2_f16(const void* x, const void* w, const void* bias, void* y, int batch, int H, int W,
                                int cin, int cout, odet_stream_t stream);
int odet_conv3x3_f16_levels(const odet_conv_level_t* levels, int num_levels, const void* w, const void* bias,
                            int batch, int cin, int cout, int relu, odet_stream_t stream);
/* The whole RpnHead (base_fpn_model.py:393-434, 188-200): the 3x3 convolution of every level as above, and in its
 * epilogue relu(conv + conv_b) (one float16 rounding) . w^T + b for the 2A score and 4A delta rows (w [6A][cout] float16,
 * b [6A] float16: rpn_score's rows, then rpn_bbox's), written as float32 into the concatenated arrays scores
 * [batch][N][2] / deltas [batch][N][4] (image strides in VALUES; the levels follow each other in the order given, H*W*A
 * anchors each).  levels[].y is not used: the cout-channel activation never goes to memory -- a workgroup walks the
 * channel tiles of its pixel slab and keeps the 32 float32 sums per pixel in registers (fixed order: deterministic; one
 * launch, no workspace).  cout = 256 or 512, 1 <= A <= 5. */
int odet_rpn_head_fused_f16(const odet_conv_level_t* levels, int num_levels, const void* conv_w, const void* conv_b,
                            const void* w, const void* b, int A, int batch, int cin, int cout, float* scores,
                            long long scores_image_stride, float* deltas, long long deltas_image_stride,
                            odet_stream_t stream);
/* A bottleneck block's 3x3 convolution AND its last 1x1 convolution in one launch (resnet_fpn.py:154-205 with the frozen
 * batch norms folded): y = relu( relu(conv3x3(x, w2) + b2) . w3^T + b3 + residual ), x NHWC float16 [batch,H,W,cin],
 * w2 [256][3][3][cin], b2 [256], w3 [n3][256], b3 [n3], residual (nullable) / y NHWC float16 [batch,H,W,n3]; the
 * 256-channel activation between the two is rounded to float16 once and never goes to memory.  cin % 64 == 0, the
 * 3x3 convolution has exactly 256 output channels (one workgroup holds them all), n3 % 64 == 0. */
int odet_conv3x3_conv1x1_f16(const void* x, const void* w2, const void* b2, const void* w3, const void* b3,
                             const void* residual, void* y, int batch, int H, int W, int cin, int n3, int relu,
                             odet_stream_t stream);
/* 1x1 convolutions and dense layers as the same LDS-staged GEMM on the matrix cores (the implicit-GEMM kernel with one
 * tap): y = relu?( x(:, ::stride, ::stride, :) . w^T + bias + residual ).  x NHWC float16 [batch,H,W,cin], w [cout][cin]
 * (a Conv2D 1x1 kernel, or a Dense kernel transposed), bias [cout] (nullable), residual (nullable) / y NHWC float16
 * [batch, ceil(H/stride), ceil(W/stride), cout].  Replaces, in the reference's ResNet-FPN (model/fpn/resnet_fpn.py): the
 * bottlenecks' first 1x1 convolution and their strided shortcut / first convolutions (:154-205; Conv2D(1x1, strides=2,
 * 'valid') reads every second pixel), the neck's P5 convolution (:339-384) and the RoI head's Dense layers (:292-336;
 * batch = 1, H = 1, W = rows).  cin % 64 == 0 and >= 128, cout % 64 == 0, stride 1 or 2.  The workgroup tile (256 / 128 /
 * 64 channels x 128..256 pixels) is picked per launch so that the workgroups fill whole rounds of the CUs. */
int odet_pointwise_f16(const void* x, const void* w, const void* bias, const void* residual, void* y, int batch,
                       int H, int W, int stride, int cin, int cout, int relu, odet_stream_t stream);
/* The end of a stage's FIRST bottleneck (resnet_fpn.py:154-205 with conv_shortcut: last 1x1 convolution + BN, convolutional
 * shortcut + BN on the block's input, Add, ReLU) as ONE contraction along the concatenated K:
 *   y = relu?( [x1 | x2(:, ::stride2, ::stride2, :)] . w^T + bias ),  w [cout][cin1 + cin2] = [w3 | w_shortcut], bias = b3 + b_sc.
 * x1 NHWC float16 [batch, ceil(H2/stride2), ceil(W2/stride2), cin1] (the block's 3x3 convolution output), x2 NHWC float16
 * [batch,H2,W2,cin2] (the block's input).  The shortcut map is never written or re-read.  cin1, cin2 % 64 == 0. */
int odet_pointwise_dual_f16(const void* x1, int cin1, const void* x2, int cin2, int H2, int W2, int stride2,
                            const void* w, const void* bias, void* y, int batch, int cout, int relu, odet_stream_t stream);
/* The network's LAST dense layer on the same kernel with float32 results: y[rows][cout] (float32) = relu?( x . w^T + bias ),
 * x [rows][cin] / w [cout][cin] float16, bias [cout] float32 (nullable) -- float32 accumulation and no rounding of the
 * result: the class logits and box regressions of the RoI head (resnet_fpn.py:327-336; a float16 logit near 10 is 0.008
 * coarse).  cout % 64 == 0: the caller pads the weight rows (Ccls + 4 Ccls -> 128) with zeros. */
int odet_dense_f16_out_f32(const void* x, const void* w, const float* bias, float* y, long long rows, int cin, int cout,
                           int relu, odet_stream_t stream);
/* A lateral 1x1 convolution of the FPN neck WITH the top-down merge in its epilogue (resnet_fpn.py:385-398):
 * y = 0.5 * resize_bilinear(top, (H, W)) + 0.5 * (x . w^T + bias), TF 1.x legacy resize (align_corners = False) in
 * float32 as odet_fpn_topdown_merge computes it, one rounding; top NHWC float16 [batch,th,tw,cout].  The lateral map
 * never goes to memory. */
int odet_lateral_merge_f16(const void* x, const void* w, const void* bias, const void* top, int th, int tw, void* y,
                           int batch, int H, int W, int cin, int cout, odet_stream_t stream);
/* The same with the 3x3 convolution's channel count as an argument: cmid = 64 (ResNet conv2), 128 (conv3) or 256 (conv4);
 * w2 [cmid][3][3][cin], b2 [cmid], w3 [n3][cmid].  odet_conv3x3_conv1x1_f16 = cmid 256. */
int odet_bottleneck_tail_f16(const void* x, const void* w2, const void* b2, const void* w3, const void* b3,
                             const void* residual, void* y, int batch, int H, int W, int cin, int cmid, int n3, int relu,
                             odet_stream_t stream);
/* The ResNet stem in one launch (resnet_fpn.py:262-289, resnet_faster_rcnn.py:31-60): ZeroPadding2D(3) -> Conv2D(64, 7x7,
 * stride 2, 'valid') + folded frozen BN -> ReLU -> ZeroPadding2D(1) -> MaxPooling2D(3x3, stride 2, 'valid'), from the
 * NHWC 3-channel image (float32: images_f16 = 0, or float16) to the NHWC float16 map [batch][PH][PW][64], PH =
 * ((H - 1) / 2 + 1 - 1) / 2 + 1.  The convolution runs on the matrix cores in float16 with float32 accumulation; neither
 * the padded image nor the convolution output go to memory.  packed_w: the weights repacked once by
 * odet_stem_pack_weights_f16 (from a float16 [64][3][7][7] tensor with the given element strides) into 64 x 7 x 32
 * float16. */
int odet_stem_pack_weights_f16(const void* w, long long stride_o, long long stride_c, long long stride_y,
                               long long stride_x, void* packed, odet_stream_t stream);
int odet_stem_conv7_pool3_f16(const void* images, int images_f16, const void* packed_w, const void* bias, void* out,
                              int batch, int H, int W, odet_stream_t stream);
/* VGG16's first convolution (vgg16_faster_rcnn.py:260-342: Conv2D(64, 3x3, padding 'same') + ReLU on the 3-channel image) in
 * one launch: NHWC image (float32: images_f16 = 0, or float16) -> NHWC float16 [batch][H][W][64], float16 products with
 * float32 accumulation on the matrix cores, + bias (+ ReLU), one rounding.  packed_w: the weights repacked once by
 * odet_conv3x3_rgb_pack_weights_f16 (from a float16 [64][3][3][3] tensor with the given element strides) into 4 x 2 x 64 x 8
 * float16. */
int odet_conv3x3_rgb_pack_weights_f16(const void* w, long long stride_o, long long stride_c, long long stride_y,
                                      long long stride_x, void* packed, odet_stream_t stream);
int odet_conv3x3_rgb_f16(const void* images, int images_f16, const void* packed_w, const void* bias, void* out,
                         int batch, int H, int W, int relu, odet_stream_t stream);
/* The float32 forms (csrc/conv_f32.hip): the detectors' PARITY mode computes in the reference's precision -- float32 x / w /
 * bias / y, exact-float32 matrix instructions (v_mfma_f32_16x16x4_f32: a chain of fmaf, no rounding the reference does
 * not have), the same tiling / staging / epilogues as the float16 entry points of the same names.  cin % 32 == 0,
 * cout % 64 == 0. */
int odet_conv3x3_f32(const void* x, const void* w, const void* bias, void* y, int batch, int H, int W,
                     int cin, int cout, int relu, odet_stream_t stream);
int odet_conv3x3_f32_levels(const odet_conv_level_t* levels, int num_levels, const void* w, const void* bias,
                            int batch, int cin, int cout, int relu, odet_stream_t stream);
/* 1x1 convolutions (stride 1 or 2) / dense layers (odet_pointwise_f16's float32 twin; cin >= 64 along K), the lateral
 * convolution with the top-down merge in its epilogue (bit-identical to odet_fpn_topdown_merge applied to the
 * convolution's float32 result) and a stage's first bottleneck's last convolution + convolutional shortcut as one
 * contraction (resnet_fpn.py:154-205, 292-336, 339-398). */
int odet_pointwise_f32(const void* x, const void* w, const void* bias, const void* residual, void* y, int batch,
                       int H, int W, int stride, int cin, int cout, int relu, odet_stream_t stream);
int odet_lateral_merge_f32(const void* x, const void* w, const void* bias, const void* top, int th, int tw, void* y,
                           int batch, int H, int W, int cin, int cout, odet_stream_t stream);
int odet_pointwise_dual_f32(const void* x1, int cin1, const void* x2, int cin2, int H2, int W2, int stride2,
                            const void* w, const void* bias, void* y, int batch, int cout, int relu, odet_stream_t stream);
/* The SPLIT-PRECISION float32 forms (csrc/conv_x3.hip): the same layers, the same float32 x / bias / y / residual / top in
 * memory and the same epilogues as the *_f32 entry points above, computed on the bfloat16 matrix instructions (16 x the
 * float32 MFMA rate): every float32 operand is the exact sum of three bfloat16 limbs (round to nearest even), a product
 * the six limb products down to 2^-16 of its size (the dropped ones are <= 2^-23 of it), float32 accumulation -- a float32
 * evaluation of the reference's float32 convolution (resnet_fpn.py:154-289, 339-407; base_fpn_model.py:393-434) within
 * float32 rounding of the float64 truth, like the fmaf chain of the *_f32 forms, not bit-identical to it.  `w3` = the
 * weight's limb planes, bfloat16 [3][cout][K] (K = taps * cin (+ cin2) in the order of the float32 weight), written ONCE per
 * weight tensor by odet_split_bf16x3 (n = cout * K float32 values -> planes [3][n]; n even).  cin (and cin2) a positive
 * multiple of 32 (the kernel's 128-byte K-steps), cout % 64 == 0.
 * `workspace` (nullable; odet_x3_workspace_bytes() bytes, 16-byte aligned, ZERO-FILLED ONCE by the caller and then only ever
 * handed to these entry points, one workspace per stream that runs them): with it a launch that would leave CUs idle (few
 * pixels, deep K: the small maps at batch 1 .. 8, the RoI head's dense layers) splits K over up to 8 workgroups per output
 * tile; the last one to finish adds the float32 parts in their fixed order (deterministic; exact on integer data) and runs the
 * epilogue.  Every launch leaves the workspace's ticket words zero again.  NULL: never split. */
size_t odet_x3_workspace_bytes(void);
int odet_split_bf16x3(const float* w, void* planes, long long n, odet_stream_t stream);
int odet_conv3x3_x3(const void* x, const void* w3, const void* bias, void* y, int batch, int H, int W,
                    int cin, int cout, int relu, void* workspace, size_t workspace_bytes, odet_stream_t stream);
int odet_conv3x3_x3_levels(const odet_conv_level_t* levels, int num_levels, const void* w3, const void* bias,
                           int batch, int cin, int cout, int relu, void* workspace, size_t workspace_bytes,
                           odet_stream_t stream);
int odet_pointwise_x3(const void* x, const void* w3, const void* bias, const void* residual, void* y, int batch,
                      int H, int W, int stride, int cin, int cout, int relu, void* workspace, size_t workspace_bytes,
                      odet_stream_t stream);
int odet_lateral_merge_x3(const void* x, const void* w3, const void* bias, const void* top, int th, int tw, void* y,
                          int batch, int H, int W, int cin, int cout, void* workspace, size_t workspace_bytes,
                          odet_stream_t stream);
int odet_pointwise_dual_x3(const void* x1, int cin1, const void* x2, int cin2, int H2, int W2, int stride2,
                           const void* w3, const void* bias, void* y, int batch, int cout, int relu,
                           void* workspace, size_t workspace_bytes, odet_stream_t stream);
/* The TWO-LIMB float16 forms of the same layers (csrc/conv_x3.hip, NL = 2): a float32 operand as h + l * 2^-11, h = f16(a),
 * l = f16((a - h) * 2^11) (round to nearest even: 11 + 11 bits and l's sign = 23 of float32's 24 bits, every operand to
 * within ONE float32 ulp), a product as h h + (h l + l h) * 2^-11 on v_mfma_f32_16x16x32_f16 (dropped: l l <= 2^-22 of it),
 * two float32 accumulators joined at the end: HALF the matrix work of the three-limb form; against float64 its error on the
 * detectors' layers is no larger than the exact-float32 form's (float32 accumulation dominates both) -- for data inside
 * float16's RANGE.  |activation| >= 65520 becomes an infinite limb: every product with it is infinite or NaN, so every sum it
 * enters is non-finite BEFORE bias / shortcut / ReLU whatever the weights' signs (never a wrong finite number), and the launch
 * REPORTS it: with a workspace, the epilogue ORs 1 into the uint32 RANGE STATUS word at byte odet_x2_status_offset() of the
 * workspace whenever a raw sum is not finite (sticky: launches only OR into it; the caller reads it after a chain of layers
 * and clears it -- it is part of the region the caller zero-fills once).  A ReLU (`v < 0 ? 0 : v` maps -inf to 0) can hide
 * the value downstream, never the flag.  Without a workspace a launch cannot report.  Activations below 2^-14
 * keep an absolute error <= 2^-36 instead of a relative one.  `w2` = float16 planes [2][cout][K] of w * 2^w_exp, written once
 * per weight tensor by odet_split_f16x2; the caller picks w_exp (|w_exp| <= 100) so that the largest |w| * 2^w_exp lies in
 * [512, 1024) -- every weight down to 2^-24 of the largest then keeps both limbs normal -- and passes the same w_exp to the
 * layer, which scales the sums back (a power of two: exact).  Everything else as the *_x3 entry points. */
int odet_split_f16x2(const float* w, void* planes, long long n, int w_exp, odet_stream_t stream);
size_t odet_x2_status_offset(void);
int odet_conv3x3_x2(const void* x, const void* w2, const void* bias, void* y, int batch, int H, int W,
                    int cin, int cout, int relu, int w_exp, void* workspace, size_t workspace_bytes, odet_stream_t stream);
int odet_conv3x3_x2_levels(const odet_conv_level_t* levels, int num_levels, const void* w2, const void* bias,
                           int batch, int cin, int cout, int relu, int w_exp, void* workspace, size_t workspace_bytes,
                           odet_stream_t stream);
int odet_pointwise_x2(const void* x, const void* w2, const void* bias, const void* residual, void* y, int batch,
                      int H, int W, int stride, int cin, int cout, int relu, int w_exp, void* workspace,
                      size_t workspace_bytes, odet_stream_t stream);
int odet_lateral_merge_x2(const void* x, const void* w2, const void* bias, const void* top, int th, int tw, void* y,
                          int batch, int H, int W, int cin, int cout, int w_exp, void* workspace, size_t workspace_bytes,
                          odet_stream_t stream);
int odet_pointwise_dual_x2(const void* x1, int cin1, const void* x2, int cin2, int H2, int W2, int stride2,
                           const void* w2, const void* bias, void* y, int batch, int cout, int relu, int w_exp,
                           void* workspace, size_t workspace_bytes, odet_stream_t stream);
/* The stem's patch matrix in float32 mode: row (image, yo, xo) = the zero-padded 7 x 7 x 3 window of conv1_pad +
 * Conv2D(64, 7x7, strides 2, 'valid') (resnet_fpn.py:262-289) in (dy, dx, channel) order, padded from 147 to 160 floats;
 * images NHWC float32 [batch,H,W,3] -> patches [batch * Ho * Wo][160], Ho = (H - 1) / 2 + 1.  The convolution is then
 * odet_pointwise_f32 with cin = 160 on it (weights [64][160] in the same order). */
int odet_stem_patches_f32(const float* images, float* patches, int batch, int H, int W, odet_stream_t stream);
/* The same for VGG16's first convolution (vgg16_faster_rcnn.py:260-342, Conv2D(64, 3x3, 'same') on the 3-channel image):
 * patches [batch][H][W][64] float32, row = the zero-padded 3 x 3 x 3 window in (dy, dx, channel) order, 27 values + zeros. */
int odet_rgb_patches3x3_f32(const float* images, float* patches, int batch, int H, int W, odet_stream_t stream);

/* 1x1 stride-1 convolution with its whole epilogue on the matrix cores (the third convolution
 * of a bottleneck block + Add([shortcut, x]) + Activation('relu'), model/fpn/resnet_fpn.py:154-205, frozen
 * BatchNormalization folded into w / bias): y[npix, cout] = relu?(x'[npix, cin] . w[cout, cin]^T + bias[cout]
 * (+ residual[npix, cout])), all float16 NHWC (w = the convolution's [cout, cin, 1, 1] weight), float32
 * accumulation, one rounding.  x' = x, or relu(x + in_bias[cin]) when in_bias != NULL (x is then the preceding
 * convolution WITHOUT its bias and ReLU: the block's 3x3 convolution; its epilogue pass disappears).
 * cin in {64, 128, 256, 512}, cout % 64 == 0, 16-byte aligned pointers; in_bias / residual may be NULL; y must not
 * alias x (it may alias residual). */
int odet_conv1x1_f16(const void* x, const void* in_bias, const void* w, const void* bias, const void* residual,
                     void* y, long long npix, int cin, int cout, int relu, odet_stream_t stream);

/* RPN head epilogue (model/fpn/base_fpn_model.py:188-200,427-432): one pyramid level's 1x1
 * convolution output level_out [B, pixels, ch] (NHWC, ch = 2A scores or 4A deltas, float32 or float16, WITHOUT
 * its bias) -> + bias[ch] -> float32 at out[b * out_image_stride + out_offset + ...]: the level's slice of the
 * concatenated [B, N, 2] / [B, N, 4] arrays the proposal stage reads (tf.reshape + tf.concat of the reference). */
int odet_rpn_pack(const void* level_out, const void* bias, long long pixels, int ch, int B, float* out,
                  long long out_image_stride, long long out_offset, int f16, odet_stream_t stream);

/* Same, for the RpnHead's two 1x1 convolutions run as ONE contraction (output channels = 2A scores then 4A
 * deltas, weights concatenated): level_out [B, pixels, 6A] + bias[6A] -> the level's slices of BOTH arrays
 * (scores [B, N, 2] at scores_offset values, deltas [B, N, 4] at deltas_offset values inside an image). */
int odet_rpn_pack_pair(const void* level_out, const void* bias, long long pixels, int A, int B, float* scores,
                       long long scores_image_stride, long long scores_offset, float* deltas,
                       long long deltas_image_stride, long long deltas_offset, int f16, odet_stream_t stream);

/* The whole RpnHead after its 3x3 convolution in one pass (base_fpn_model.py:393-434, 188-200):
 * conv_out [B, pixels, 512] = the 3x3 convolution WITHOUT its bias (float16 NHWC), conv_bias [512];
 * t = relu(conv_out + conv_bias); scores = t . w[0:2A]^T + bias[0:2A]; deltas = t . w[2A:6A]^T + bias[2A:6A]
 * (w [6A, 512] float16 = rpn_score's rows then rpn_bbox's, bias [6A] float16), written as float32 into the level's
 * slices of the concatenated [B, N, 2] / [B, N, 4] arrays (offsets / strides in VALUES, as odet_rpn_pack_pair).
 * Matrix cores, float32 accumulation; 1 <= A <= 4. */
int odet_rpn_head_tail_f16(const void* conv_out, const void* conv_bias, const void* w, const void* bias,
                           long long pixels, int A, int B, float* scores, long long scores_image_stride,
                           long long scores_offset, float* deltas, long long deltas_image_stride,
                           long long deltas_offset, odet_stream_t stream);

/* ---- multi-GPU detection records ------------------------------------------------------ */

/* Native addition (the reference has no multi-GPU path): packs the padded post-ops outputs of
 * one image into the fixed-size record exchanged by the image-parallel all-gather:
 * out_record float32 [max_det*6 + 1] = max_det rows (x1,y1,x2,y2,score,label), padded rows
 * zero with score -1, then the count.  capacity = rows available in boxes/labels/scores. */
int odet_pack_detections(const float* boxes, const int32_t* labels, const float* scores,
                         const int32_t* count_dev, int capacity, int max_det, float* out_record,
                         odet_stream_t stream);

/* ---- whole-step descriptor + native executor (throughput serving) ---------------------- */

/* One image through the FPN hot path as the reference's BaseFPN.call runs it
 * (model/fpn/base_fpn_model.py:208-276 minus the dense conv parts): every buffer is caller-owned
 * device memory, the struct is plain data.  odet_fpn_step_enqueue() issues the selected stages on
 * `stream` from the calling thread:
 *   ODET_STAGE_PROPOSALS = odet_fpn_proposals   (anchors, fg softmax, RegionProposal, _assign_levels; also the
 *                          processing order into roi_order when num_proposals <= ODET_FUSED_ORDER_MAX_ROIS)
 *   ODET_STAGE_ROI       = odet_roi_pool        (RoiPoolingCropAndResize2 over maps[0..num_maps); odet_roi_order
 *                          first when num_proposals > ODET_FUSED_ORDER_MAX_ROIS)
 *   ODET_STAGE_DETECT    = odet_post_ops_record (post_ops_prediction + detection record) */
#define ODET_STAGE_PROPOSALS 1
#define ODET_STAGE_ROI 2
#define ODET_STAGE_DETECT 4
#define ODET_STAGE_ALL 7

typedef struct {
  /* proposals */
  int32_t image_h, image_w;
  int32_t num_levels, A;                         /* RPN pyramid levels / anchors per cell */
  int32_t fh[ODET_MAX_LEVELS], fw[ODET_MAX_LEVELS], stride[ODET_MAX_LEVELS];
  float wh[ODET_MAX_LEVELS * ODET_MAX_ANCHORS_PER_CELL * 2];
  float rpn_means[4], rpn_stds[4];
  int32_t num_proposals;
  float rpn_nms_iou;
  int32_t min_level, max_level, blind_chunks;
  int32_t nms_first_chunk;   /* 0 = auto (~1.5 x num_proposals candidates); else candidates of the first NMS chunk,
                                <= 4096: a wider chunk lets heavy suppression (trained-like score clusters) finish
                                inside the one sync-free chunk of batched launches */
  /* roi */
  int32_t num_maps, channels, pool_size;
  int32_t maps_f16;          /* != 0: float16 feature maps and float16 roi_features (odet_roi_pool_f16) */
  odet_level_t maps[ODET_MAX_LEVELS];
  /* detect */
  int32_t ccls, num_classes, max_per_class, max_per_image;
  float roi_means[4], roi_stds[4];
  float nms_iou, score_threshold, min_edge;
  /* inputs (device) */
  const float* rpn_logits;   /* [n,2] */
  const float* rpn_deltas;   /* [n,4] */
  const float* cls_scores;   /* [num_proposals, ccls] */
  const float* cls_deltas;   /* [num_proposals, ccls, 4] */
  /* outputs / scratch (device, caller-owned) */
  float* rois; int32_t* roi_idx; int32_t* roi_count; int32_t* nms_done;
  float* sorted_rois; int32_t* roi_level; int64_t* roi_perm; int32_t* level_counts;
  void* roi_features;        /* [num_proposals, pool, pool, channels] float32 (float16 when maps_f16) */
  int32_t* roi_order;        /* nullable scratch int32 [num_proposals]: spatial processing order (odet_roi_order) */
  float* det_boxes; int32_t* det_labels; float* det_scores; int32_t* det_count; float* record;
  void* ws_rpn; size_t ws_rpn_bytes;
  void* ws_post; size_t ws_post_bytes;
  odet_stream_t stream;
  /* profiling (nullable): HIP events (odet_prof_event_create) attached to the RoI dispatch of this step -- in a
   * batch those of the first step bracket the one launch all its images share */
  void* roi_start_event; void* roi_stop_event;
  /* != 0: the caller promises that ws_rpn was zero-filled once after its allocation and has only ever been handed to
   * this library since: every call leaves the proposal stage's header clean again, so no launch / memset is spent on
   * zeroing it.  (ws_post has its own promise, ws_post_clean, since version 101.) */
  int32_t ws_rpn_clean;
  /* != 0: a single-level Faster R-CNN step (model/faster_rcnn/base_faster_rcnn_model.py:126-198 minus the dense
   * parts) instead of an FPN step: num_levels = num_maps = 1; fh[0] x fw[0] cells of stride[0] with A anchors each,
   * wh[0 .. 4A) = the anchor base (generate_anchor_base as float32 rows x1,y1,x2,y2); rpn_logits [fh*fw, 2A] in the
   * [A bg | A fg] layout; no level assignment (sorted_rois must alias rois; roi_level, roi_perm, level_counts NULL);
   * RoI crops normalised by maps[0].stride (ODET_ROI_NORM_STRIDE) with roi_pool_mode; min_edge = the stride. */
  int32_t single_level;
  int32_t roi_pool_mode;     /* single_level: ODET_ROI_POOL_MAX2 (VGG16) | ODET_ROI_POOL_NONE (ResNet C4); FPN: MAX2 */
  /* != 0: the same promise for ws_post (zero-filled once, only ever handed to this library): the post-processing
   * launch keeps a wrapping ticket counter at its end, which every COMPLETED call leaves at 0.  With 0 here the library
   * issues the 4-byte memset on the stream before the launch (the default; costs nothing at graph replay).  After a
   * failed call (an error from odet_exec_wait / a launch error) zero-fill ws_post again before promising this. */
  int32_t ws_post_clean;
} odet_fpn_step_t;

size_t odet_fpn_step_sizeof(void);
int odet_fpn_step_enqueue(const odet_fpn_step_t* step, int stages);
/* `count` (<= ODET_MAX_STEP_BATCH) images whose steps agree in every shape, parameter and the stream,
 * processed by the SAME kernel launches (one grid dimension = image): per-launch costs are paid once
 * per batch and the single-workgroup stages of the images run side by side.  Needs the sync-free NMS
 * mode (nms_done != NULL); blind_chunks >= 1 (chunk 1 comes from the same ranked selection in launches
 * shared by the batch, chunks 2.. per image on the full order).  An image whose NMS did not complete
 * inside its chunks reports zero proposals and nms_done = 0 (see odet_nms). */
#define ODET_MAX_STEP_BATCH 8
int odet_fpn_step_enqueue_batch(const odet_fpn_step_t* const* steps, int count, int stages);

/* Native executor: `num_workers` host threads, each draining its own FIFO of (step, stages) jobs by
 * calling odet_fpn_step_enqueue.  A HIP kernel launch costs ~3 us of host time and one image is ~11
 * launches, so one host thread saturates near 20k images/s; with one worker per stream the images of
 * different streams are enqueued in parallel (the reference has no counterpart: it is a
 * single-threaded Python loop).  A job only enqueues GPU work; completion is the stream's business.
 * The step structs must stay alive and unchanged until odet_exec_wait() has returned.  Work that
 * must be ordered after a job on the same stream from another thread (e.g. a torch op) has to be
 * issued after odet_exec_wait(). */
typedef struct odet_exec odet_exec_t;
odet_exec_t* odet_exec_create(int num_workers);
void odet_exec_destroy(odet_exec_t* ex);
int odet_exec_submit(odet_exec_t* ex, int worker, const odet_fpn_step_t* step, int stages);
int odet_exec_submit_batch(odet_exec_t* ex, int worker, const odet_fpn_step_t* const* steps, int count,
                           int stages);
/* blocks until every submitted job has been enqueued; returns the first error any job hit (0 = none;
 * text through odet_exec_last_error) and clears it */
int odet_exec_wait(odet_exec_t* ex);
const char* odet_exec_last_error(odet_exec_t* ex);

#ifdef __cplusplus
}
#endif
#endif /* ODET_H_ */

#!/usr/bin/env python3
"""bench.py -- images/sec of the MI355X-native Faster-R-CNN/FPN detection hot path.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json metric "images/sec ResNet-101-FPN @ 800x1333", configs[2]): every image goes
through region_proposal -> roi_pooling -> prediction (+ anchor generation, fg softmax, level assignment)
at the ResNet-101-FPN shapes: 267 069 anchors, 1000 proposals, P2..P5 x 256 channels, 21 classes.  The
dense conv parts of the model (backbone, neck, RPN head, RoI head) are NOT part of this path (SURVEY.md
section 8): their outputs are the synthetic inputs, resident in HBM before the timed region.

A STEP = `--rounds-per-step` (48) rounds over the `streams` x `batch` (3 x 8 = 24) in-flight image slots of a
GPU = 1152 images per GPU (`config.images_per_step_per_gpu`), so the driver's `--steps 20 --warmup 5` times
23 040 images (~0.75 s), not 20.

Serving arrangement: images are independent, so `--streams` HIP streams (each fed by a native enqueue
thread of the library) carry `--batch` images each whose kernels share their launches (one grid dimension
= image).  One process per GPU; images shard by rank (weak scaling: the same number of images per GPU per
step); with N > 1 every round of streams x batch images ends with ONE RCCL all-gather of the fixed-size
detection records.

Prints ONE JSON line (rank 0), <= 8 KB (LINE_LIMIT; asserted in tests/test_bench_contract.py), with the driver's contract plus
  `roofline`     dominant kernel = the fused RoI crop+pool kernel (HBM bound), the 8-image launch of one stream
                 group: >= 10 launches in a short phase AFTER the timed region, each alone on the GPU with cold
                 maps, HIP events attached to the dispatch.  Those launches run under their own kernel name
                 (k_roi_pool<..., 1>, same code), so the kernel-stats summary of a profiled run of this command
                 (profiles/) has a row whose average IS `kernel_ms`.  `frac` prices SURVEY 8(d)'s algorithmic
                 bytes B_roi (no credit for reuse between RoIs); `hbm_frac_measured` prices the HBM bytes the PMC
                 counters saw (profiles/roi_pool_traffic.json); `B_min` / `B_taps` bracket B_roi; `calibration` = a kernel that
                 only moves B_min bytes with the RoI kernel's instructions (odet_calib_stream_mix), timed under the same
                 protocol right after: what this box's memory system needs for the launch's read : write mix.
  `cpu_baseline` the C restatement of the reference path timed on this box's host cores (kind "port"), all cores
                 (`value`) and one thread (`value_1thread`); it also carries the mAP delta of the evaluation loop.
  `value_clustered` the same path, same protocol, on trained-like (clustered) RPN scores: a second timed region; the
                 sync-free NMS plan is widened by itself, rung by rung, when a distribution needs it (`config.replanned`).
  `multi_rank`   what a 2 / 4 / 8-GPU run needs to validate itself: the backend's world size (RCCL), per-rank img/s min / max,
                 the number of all-gathers issued and their record bytes.
  `detail`       (N = 1) where the secondary records are and how their process ended (below).
  `summary`      LAST key, <= 1 KB, values only: every headline figure of this line AND of the side file.

Everything that is not the headline -- (N = 1 only) -- runs AFTER the line's own measurements in a fresh CHILD process
(`bench.py --detail-child`, started with subprocess, own session, hard timeout; never an exec of the process that holds the
GPU) and writes `bench_detail.json` next to this file, rewritten after every leg (progress on stderr).  A hang or GPU fault in
any of those legs costs that leg: the parent prints its line regardless.  The side file holds
  `config5`      BASELINE configs[4] as a measured record: the same hot path at 1333x1333 (446 118 anchors, 1000
                 proposals), 81 classes, caps 100 per class / 300 per image (config/faster_rcnn_config.py:93-113's COCO caps),
                 float16 feature maps into the RoI kernel -- throughput, its RoI launch alone, its roofline fractions.
  `e2e`          a second, separately labelled record: the assembled detectors end to end (hand-written
                 convolutions around the hot path, no library convolution or GEMM): ResNet-101-FPN fp32 = the reference's
                 precision (exact / three-limb / two-limb forms), fp16 = throughput mode (narrower than the reference; eager and
                 as one HIP graph), ResNet-50 C4 and VGG16 (BASELINE configs 2 and 1), and the accuracy gates
                 `map_delta_vs_fp32` (a mode vs the exact-float32 detector on identical weights and annotated synthetic
                 scenes, the reference's evaluation loop; `ci_inside_bar` = both ends of the paired-bootstrap CI95 within +-0.002).
The whole default run aims at `--time-budget` (190 s): the gates run on as many scenes as fit.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

IMAGE_SHAPE = (800, 1333)
NUM_CLASSES = 21
NUM_PROPOSALS = 1000
CHANNELS = 256
HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s)


def algorithmic_roi_bytes(sorted_rois, levels, level_shapes, image_shape, channels, pool=7, elem=4):
    """B_roi = sum_r U_r*C*s + R*P*P*C*s + R*16, U_r = unique feature cells tapped by
    RoI r on its level (first/last in-bounds sample coordinate of the 2P x 2P grid, FPN normalisation
    of model/roi_pooling.py:30-35 + TF crop_and_resize); B_min = sum_k min(sum_{r in k} U_r, H_k*W_k)*C*s + out
    + rois (perfect reuse between RoIs); B_taps = R*crop^2*4*C*s + out (every tap fetched)."""
    crop = 2 * pool
    H_img, W_img = np.float32(image_shape[0]), np.float32(image_shape[1])
    total_cells = 0
    per_level = {}
    for r, l in zip(sorted_rois, levels):
        Hk, Wk = level_shapes[int(l)]
        spans = []
        for lo, hi, img, dim in ((r[1], r[3], H_img, Hk), (r[0], r[2], W_img, Wk)):
            lo_n, hi_n = np.float32(lo) / img, np.float32(hi) / img
            lim = np.float32(dim - 1)
            scale = (hi_n - lo_n) * lim / np.float32(crop - 1)
            coords = lo_n * lim + np.arange(crop, dtype=np.float32) * scale
            ok = coords[(coords >= 0) & (coords <= lim)]
            if ok.size == 0:
                spans.append(0)
            else:
                first, last = ok.min(), ok.max()
                spans.append(int(min(np.ceil(last), dim - 1) - max(np.floor(first), 0) + 1))
        total_cells += spans[0] * spans[1]
        per_level[int(l)] = per_level.get(int(l), 0) + spans[0] * spans[1]
    R = len(sorted_rois)
    out_bytes = R * pool * pool * channels * elem
    min_cells = sum(min(c, level_shapes[l][0] * level_shapes[l][1]) for l, c in per_level.items())
    return dict(B_roi=total_cells * channels * elem + out_bytes + R * 16,
                B_min=min_cells * channels * elem + out_bytes + R * 16,
                B_taps=R * crop * crop * 4 * channels * elem + out_bytes, out=out_bytes, unique_cells=total_cells)


def _cpu_leg(host, image_shape, threads, budget_s, max_images):
    """images of the workload through the C restatement on `threads` OpenMP threads -> per-image seconds"""
    from oracle import c_oracle as co
    K = NUM_PROPOSALS
    t_all = []
    t_start = time.perf_counter()
    scratch = np.empty((K, 14, 14, CHANNELS), np.float32)
    while len(t_all) < max_images and (time.perf_counter() - t_start) < budget_s:
        t0 = time.perf_counter()
        anchors = co.fpn_anchors(image_shape)
        fg = co.rpn_fg_fpn(host['rpn_logits'])
        rois, idx = co.region_proposal(host['rpn_deltas'], anchors, fg, image_shape, K, 0.7)
        lv, perm, cnt = co.assign_levels(rois)
        srois = rois[perm]
        slv = lv[perm]
        for l in range(4):
            sel = srois[slv == l + 2]
            if sel.shape[0]:
                co.roi_pool(host['feats'][l], sel, image_shape=image_shape, pool=7, threads=threads,
                            scratch=scratch[:sel.shape[0]])
        k = rois.shape[0]
        co.post_ops(host['cls_scores'][:k], host['cls_deltas'][:k], srois, image_shape, [0, 0, 0, 0],
                    [.1, .1, .2, .2], 50, 50, 0.3, 0.0, 16, NUM_CLASSES)
        t_all.append(time.perf_counter() - t0)
    return t_all


def cpu_baseline(host, image_shape, budget_s=8.0, max_images=24):
    """The reference path as the TF-eager CPU code runs it (C restatement, oracle/oracle.c): heap NMS
    over all anchors, un-fused 14x14 crop -> max-pool, sequential per-class loop.  Bounded sample, in both thread modes
    of BASELINE.md section 3: many threads (what TF's intra-op pool would use; capped at the PHYSICAL cores -- the 128
    hardware threads of these hosts are shared with other tenants and the OpenMP crops lost to their own fork / join) and
    one thread.  `value` quotes the FASTER leg (min-of-N per leg is reported beside the medians)."""
    from oracle import c_oracle as co
    logical = os.cpu_count() or 1
    threads = max(1, min(logical // 2 if logical >= 4 else logical, co.max_threads(), 64))
    _cpu_leg(host, image_shape, threads, 4.0, 1)                               # (page-in / OpenMP pool start-up)
    t_all = _cpu_leg(host, image_shape, threads, budget_s, max_images)
    t_one = _cpu_leg(host, image_shape, 1, budget_s, max_images)
    t, t1 = float(np.median(t_all)), float(np.median(t_one))
    md = map_delta_vs_port()
    fast_threads = threads if t <= t1 else 1
    return dict(map_delta=md, value=1.0 / min(t, t1), unit='img/s', cores=fast_threads, kind='port',
                value_threads=1.0 / t, threads=threads, value_1thread=1.0 / t1,
                ms_per_image=min(t, t1) * 1e3, ms_per_image_threads=t * 1e3, ms_per_image_1thread=t1 * 1e3,
                ms_per_image_min_threads=float(np.min(t_all)) * 1e3, ms_per_image_min_1thread=float(np.min(t_one)) * 1e3,
                samples=len(t_all), samples_1thread=len(t_one), logical_cpus=logical,
                sample='%d (%d threads) + %d (one thread) images of the same 800x1333 FPN hot-path workload, medians, the faster '
                       'leg quoted; C restatement of the reference path (heap NMS over 267069 anchors single-threaded, un-fused '
                       'crop 14x14 + max-pool on OpenMP threads, sequential class loop)' % (len(t_all), threads, len(t_one)))


def map_delta_vs_port(num_images=16):
    """BASELINE metric's second half, 'mAP delta vs ref': the evaluation loop of
    evaluation/pascal_eval_files_utils.py:76-106 on identical synthetic im_detect outputs, once on the
    GPU (odet_eval_detect) and once through the restated reference loop (oracle, checker only), both
    scored with VOC07 11-point AP (evaluation/detectron_pascal_evaluation_utils.py)."""
    from oracle import oracle_np as on
    from tf_eager_object_detection_amd import synthetic as syn
    from tf_eager_object_detection_amd.evaluation import pascal_eval as pe
    rng = np.random.default_rng(2024)
    shapes = [(375, 500), (500, 375), (333, 500), (480, 640)]
    kw = dict(score_threshold=0.05, iou_threshold=0.5, max_objects_per_class=50, max_objects_per_image=50, min_size=10)
    dg, dr, gb, gl = [], [], [], []
    for i in range(num_images):
        im = syn.eval_image(rng, raw_shape=shapes[i % 4], num_rois=300)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
        dg.append(pe.detect_image(t(im['scores']), t(im['deltas']), t(im['rois']), im['img_scale'], im['raw_h'],
                                  im['raw_w'], **kw))
        dr.append(on.eval_detect_image(im['scores'], im['deltas'], im['rois'], im['img_scale'], im['raw_h'],
                                       im['raw_w'], **kw))
        gb.append(im['gt_boxes'])
        gl.append(im['gt_labels'])
    m_gpu = pe.evaluate_detections(dg, gb, gl, use_07_metric=True)[0]
    m_ref = pe.evaluate_detections(dr, gb, gl, use_07_metric=True)[0]
    return dict(map_gpu=m_gpu, map_port=m_ref, delta=m_gpu - m_ref, images=num_images, metric='VOC07 11-point mAP',
                data='synthetic im_detect outputs (tf_eager_object_detection_amd.synthetic.eval_image)')


def load_traffic(workload_key, images_per_launch):
    """HBM bytes per launch of the RoI kernel from the committed rocprofv3 --pmc run (profiles/), scaled to
    the images of the timed launch when the PMC run used another launch shape."""
    p = os.path.join(ROOT, 'profiles', 'roi_pool_traffic.json')
    try:
        d = json.load(open(p))
        for rec in [d] + list(d.get('also', [])):             # (one record per workload: float32 maps, float16 maps)
            if rec.get('workload') == workload_key:
                # provenance: the counters need their own rocprofv3 --pmc passes, so this figure is NOT from the run that
                # prints it -- a reader sees the file, the round / box it was collected on and how
                src = {'measured_in_this_run': False, 'file': 'profiles/roi_pool_traffic.json',
                       'collected': d.get('collected', d.get('note')), 'corrections': rec.get('corrections')}
                return rec.get('hbm_bytes_per_launch') / float(rec.get('images_per_launch', 1)) * images_per_launch, src
    except Exception:
        pass
    return None, None


def e2e_record(dtype_name, batch, budget_s=6.0, family='fpn', graph=False, eager=True, f32_form='exact'):
    """An assembled detector end to end on synthetic images (random-init weights): backbone (+ neck) + RPN head + hot
    path + RoI head + post-ops; not the headline metric (that one is the hot path): a second, separately labelled
    record.  family = 'fpn': ResNet-101-FPN @ 800x1333 (BASELINE config 3); 'c4': ResNet-50 C4 Faster R-CNN @ 800x1333
    (config 2); 'vgg16': VGG16 Faster R-CNN @ 600x800 (config 1).  fp32 = the reference's precision (parity mode).
    graph: also replay the pass as ONE HIP graph (what a batch-1 latency step needs: ~140 launches = one host call)."""
    from tf_eager_object_detection_amd.model.fpn_detector import ResNetFpnDetector
    from tf_eager_object_detection_amd.model.frcnn_detector import ResNetC4Detector, Vgg16Detector
    dt = {'fp32': torch.float32, 'fp16': torch.float16}[dtype_name]
    torch.manual_seed(0)
    image_shape = (600, 800) if family == 'vgg16' else IMAGE_SHAPE
    if family == 'fpn':
        model = ResNetFpnDetector(101, NUM_CLASSES, image_shape, NUM_PROPOSALS, dtype=dt, max_batch=batch,
                                  blind_chunks=2, batched=True, f32_form=f32_form).prepare()
    elif family == 'c4':
        model = ResNetC4Detector(50, NUM_CLASSES, image_shape, 300, dtype=dt, max_batch=batch, blind_chunks=4, f32_form=f32_form).prepare()
    else:
        model = Vgg16Detector(NUM_CLASSES, image_shape, 300, dtype=dt, max_batch=batch, blind_chunks=4, f32_form=f32_form).prepare()
    rng = np.random.default_rng(0)
    img = (rng.uniform(0, 255, (batch,) + image_shape + (3,)) - np.float32([103.939, 116.779, 123.68])).astype(np.float32)
    img = torch.from_numpy(img).cuda()
    t0 = time.perf_counter()
    for _ in range(3):
        out = model(img)
    torch.cuda.synchronize()
    warm_s = time.perf_counter() - t0
    rec = dict(unit='img/s', batch=batch, dtype=dtype_name if f32_form == 'exact' else dtype_name + ' split precision (%s)' % f32_form,
               model={'fpn': 'ResNet-101-FPN', 'c4': 'ResNet-50 C4 Faster R-CNN', 'vgg16': 'VGG16 Faster R-CNN'}[family],
               image=list(image_shape), weights='random init', data='synthetic', warmup_s=warm_s)
    if eager:
        steps, t0 = 0, time.perf_counter()
        while True:
            out = model(img)
            steps += 1
            if steps >= 5 and (steps % 5) == 0:
                torch.cuda.synchronize()
                if time.perf_counter() - t0 > budget_s or steps >= 400:
                    break
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        rec.update(value=steps * batch / el, steps=steps, ms_per_image=el / (steps * batch) * 1e3)
    rec['nms_done'] = int(all(int(v) == 1 for v in model._steps.nms_done_all[:batch].tolist()))
    rec['detections_image0'] = int(out[0][3].item())
    if f32_form == 'x2':                                # (passes whose activations left float16's range and were repeated on three limbs)
        rec['range_reruns'] = int(model.range_reruns)
    if graph:
        try:
            run = model.capture(batch)
            for _ in range(3):
                run(img)
            torch.cuda.synchronize()
            n_g, t0 = 0, time.perf_counter()
            while n_g < 2000:
                run(img)
                n_g += 1
                if n_g % 10 == 0:
                    torch.cuda.synchronize()
                    if time.perf_counter() - t0 > min(budget_s, 4.0):
                        break
            torch.cuda.synchronize()
            rec['value_hip_graph'] = n_g * batch / (time.perf_counter() - t0)
            rec['ms_per_pass_hip_graph'] = 1e3 * batch / rec['value_hip_graph']
        except Exception as ex:
            rec['value_hip_graph'] = 'capture failed: %s' % ex
        if not eager:
            rec['value'] = rec['value_hip_graph']
    del model
    torch.cuda.empty_cache()
    return rec


E2E_CONV_PATH = ('hand-written HIP kernels only, at every batch size (float16: 3x3 implicit GEMM incl. the fused RpnHead and '
                 'bottleneck tails and its 64 x 64 ring form for the small maps at batch 1-2, the pointwise GEMM form for the 1x1 / '
                 'strided / dense layers and the laterals with the top-down merge in their epilogue, the register-resident 1x1 '
                 'kernel, the fused stem; float32: the same forms on exact-float32 matrix instructions) -- the detectors have '
                 'no library convolution / GEMM route (model/fpn_detector.py)')


HARD_LIMIT_EXTRA_S = 25.0   # the child plans inside its budget; this much later a hang is assumed
LINE_LIMIT = 8192          # bytes of the ONE stdout line (the driver's parser lost a 22 KB line in round 5; 15 KB parsed)

_RENDEZVOUS_ENV = ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'LOCAL_WORLD_SIZE', 'GROUP_RANK', 'GROUP_WORLD_SIZE', 'ROLE_RANK',
                   'ROLE_WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'TORCHELASTIC_RUN_ID')


class _Terminated(Exception):
    pass


def run_detail_child(args, t_main):
    """Start `bench.py --detail-child <side file>` as a fresh CHILD process (own session; stdout -> this process's stderr),
    wait for it under a hard limit, and return what it left in the side file (+ how it ended).  The child plans its legs
    inside the budget it is given; the hard limit (budget + HARD_LIMIT_EXTRA_S) and SIGTERM to this process only catch a hang: the child's
    process group -- the one started here, by pid -- is killed, and the side file still holds every finished leg."""
    import signal
    import subprocess
    path = os.path.abspath(args.detail_out)
    for q in (path, path + '.tmp'):
        try:
            os.remove(q)
        except OSError:
            pass
    budget = max(args.time_budget - (time.perf_counter() - t_main) - 5.0, 1.0)
    cmd = [sys.executable, os.path.abspath(__file__), '--detail-child', path, '--time-budget', '%.1f' % budget,
           '--steps', str(args.steps), '--warmup', str(args.warmup), '--rounds-per-step', str(args.rounds_per_step),
           '--streams', str(args.streams), '--batch', str(args.batch), '--blind-chunks', str(args.blind_chunks),
           '--nms-first-chunk', str(args.nms_first_chunk), '--roofline-samples', str(args.roofline_samples),
           '--gate-images', str(args.gate_images)]
    cmd += ['--no-e2e'] if args.no_e2e else []
    cmd += ['--no-config5'] if args.no_config5 else []
    cmd += ['--trace'] if args.trace else []
    env = {k: v for k, v in os.environ.items() if k not in _RENDEZVOUS_ENV}
    t0 = time.perf_counter()
    try:
        proc = subprocess.Popen(cmd, stdout=sys.stderr, stderr=sys.stderr, env=env, start_new_session=True)
    except Exception as ex:                               # (a box that refuses the child: the line does not depend on it)
        return {'error': 'could not start the detail process: %s: %s' % (type(ex).__name__, ex), 'file': None, 'complete': False,
                'child_rc': None, 'child_s': 0.0, 'legs_done_n': 0}

    def on_term(signum, frame):
        raise _Terminated()
    try:
        old = signal.signal(signal.SIGTERM, on_term)
    except (ValueError, OSError):                         # (not the main thread)
        old = False
    try:
        rc = proc.wait(timeout=budget + HARD_LIMIT_EXTRA_S)
    except (subprocess.TimeoutExpired, _Terminated, KeyboardInterrupt) as ex:
        rc = 'killed (%s)' % type(ex).__name__
        try:
            os.killpg(proc.pid, signal.SIGKILL)          # exactly the group started above
        except OSError:
            pass
        try:
            proc.wait(timeout=10.0)
        except Exception:
            pass
    finally:
        # (a handler installed from C -- a profiler's -- reads back as None: nothing of Python's to restore then)
        if old is not False:
            try:
                signal.signal(signal.SIGTERM, old if old is not None else signal.SIG_DFL)
            except (TypeError, ValueError, OSError):
                pass
    detail = {}
    try:
        detail = json.load(open(path))
    except Exception as ex:
        detail = {'error': 'no side file: %s: %s' % (type(ex).__name__, ex)}
    detail.update(file=os.path.relpath(path, ROOT) if path.startswith(ROOT) else path, child_rc=rc,
                  child_s=round(time.perf_counter() - t0, 1), legs_done_n=len(detail.get('legs_done', [])))
    detail.setdefault('complete', False)
    try:                                                  # (the side file also says how its writer ended)
        with open(path + '.tmp', 'w') as f:
            json.dump(detail, f)
        os.replace(path + '.tmp', path)
    except OSError:
        pass
    return detail


def summarise_detail(summary, detail):
    """the headline figures of the side file's records -> `summary` (values only; the records themselves stay in the file)"""
    e2e = detail.get('e2e') or {}
    c5 = detail.get('config5')

    def rate(name, key='value'):
        v = (e2e.get(name) or {}).get(key)
        return round(v, 1) if isinstance(v, (int, float)) else None
    if isinstance(c5, dict) and 'value' in c5:
        # [img/s, us of the 8-image RoI launch, its fraction of 8 TB/s on B_min]
        summary['cfg5_hot_path'] = [round(c5['value'], 1), round(c5['roofline']['kernel_ms'] * 1e3, 1), round(c5['roofline']['frac_on_B_min'], 3)]
    elif c5 is not None:
        summary['cfg5_hot_path'] = 'error'
    if len(e2e) > 1:
        summary.update(e2e_fp16=[rate('fp16'), 60], e2e_fp16_b1_graph=rate('fp16_b1', 'value_hip_graph'),
                       e2e_fp16_b1_eager=rate('fp16_b1'), e2e_fp16_b4=[rate('fp16_b4'), rate('fp16_b4', 'value_hip_graph')],
                       e2e_fp16_b8=[rate('fp16_b8'), rate('fp16_b8', 'value_hip_graph')], e2e_fp32=[rate('fp32'), 30],
                       e2e_fp32_x3=[rate('fp32_x3'), 30], e2e_fp32_x3_b1_graph=rate('fp32_x3_b1', 'value_hip_graph'),
                       fp32_x2=[rate('fp32_x2'), rate('fp32_x2_b1', 'value_hip_graph')],   # (+ mAP delta, kept-anchor agreement)
                       c4_fp16=[rate('fp16_resnet50_c4'), 60], vgg16_fp16=[rate('fp16_vgg16_600x800'), 64],
                       c4_vgg16_fp32_x3=[rate('fp32_x3_resnet50_c4'), rate('fp32_x3_vgg16_600x800')])
        for name, key in (('fp16', 'fpn'), ('fp16_resnet50_c4', 'c4'), ('fp16_vgg16_600x800', 'vgg16'), ('fp32_x3', 'x3'), ('fp32_x2', 'x2')):
            g = (e2e.get(name) or {}).get('map_delta_vs_fp32') or e2e.get(name + '_map_delta_vs_fp32')
            if not isinstance(g, dict) or 'map_delta' not in g:
                continue
            lo, hi = g['map_delta_ci95_paired_bootstrap']
            if key == 'x2':         # (one compact entry: img/s at 30, batch-1 graph img/s, mAP delta, kept-anchor agreement)
                summary['fp32_x2'] = summary['fp32_x2'][:2] + [round(g['map_delta'], 4), round(g['rpn_kept_index_agreement_mean'], 4)]
                continue
            # [mAP delta, CI95 low, CI95 high, scenes, both ends of the interval within +-0.002 (1 / 0)]
            summary['map_delta_' + key] = [round(g['map_delta'], 4), round(lo, 4), round(hi, 4), g['images'], int(bool(g['ci_inside_bar']))]
            if key == 'x3':
                summary['x3_vs_exact'] = [round(g['rpn_kept_index_agreement_mean'], 4), float('%.1e' % g['p99_abs_dscore'])]
    summary['detail'] = [detail.get('file'), int(bool(detail.get('complete'))), detail.get('legs_done_n')]


def compact_line(result):
    """json.dumps(result) kept <= LINE_LIMIT bytes: the contract keys, `config`, `roofline`, `cpu_baseline`, `multi_rank`,
    `summary` stay; should the line ever outgrow the limit, its explanatory strings go first (recorded in `dropped`)."""
    line = json.dumps(result)
    dropped = []
    for path_ in (('cpu_baseline', 'sample_detail'), ('roofline', 'calibration', 'kernel'), ('roofline', 'traffic_source'),
                  ('config', 'second_distribution'), ('cpu_baseline', 'map_delta', 'data'), ('roofline', 'kernel'),
                  ('multi_rank', 'collective'), ('config', 'step')):
        if len(line) <= LINE_LIMIT:
            break
        d = result
        for k in path_[:-1]:
            d = d.get(k) if isinstance(d, dict) else None
        if isinstance(d, dict) and path_[-1] in d:
            del d[path_[-1]]
            dropped.append('.'.join(path_))
            summ = result.pop('summary', None)
            result['dropped'] = dropped
            if summ is not None:
                result['summary'] = summ                  # (stays the LAST key)
            line = json.dumps(result)
    return line



def self_launch(n, argv):
    """`python bench.py --gpus N` without a launcher's environment: start `python -m torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py <same arguments>` as a CHILD process (never an
    exec, and before this process made any GPU call), pass its stdout (rank 0's one JSON line) and stderr through, and
    return its exit code.  The torchrun form of the driver keeps working: it sets WORLD_SIZE and never reaches this."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def main():
    t_main = time.perf_counter()
    # (before anything initialises HSA: dmabuf IPC is the only kind the host driver supports)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--rounds-per-step', type=int, default=48,
                    help='a step = this many rounds over the streams x batch in-flight image slots')
    ap.add_argument('--scores', choices=['distinct', 'clustered'], default='distinct',
                    help='RPN score distribution (SURVEY 8d: distinct; clustered = trained-like)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-e2e', action='store_true', help='skip the end-to-end detector record')
    ap.add_argument('--no-config5', action='store_true', help='skip the BASELINE config-5 record (1333x1333, 81 classes, fp16 maps)')
    ap.add_argument('--no-second-distribution', action='store_true',
                    help='skip the second timed region (the other RPN score distribution: value_clustered)')
    ap.add_argument('--gate-images', type=int, default=4096,
                    help='held-out scenes of the float16-vs-float32 mAP gate of the ResNet-101-FPN detector (e2e); the C4 / VGG16 '
                         'gates run a quarter of it')
    ap.add_argument('--streams', type=int, default=3, help='HIP streams (stream groups) per GPU')
    ap.add_argument('--batch', type=int, default=8, help='images that share the kernel launches of a stream (1..8)')
    ap.add_argument('--blind-chunks', type=int, default=1, help='NMS chunks enqueued without a host check')
    ap.add_argument('--nms-first-chunk', type=int, default=0,
                    help='candidates of the first (sync-free) NMS chunk, 0 = auto ~1.5K; 4096 for score distributions with '
                         'heavy suppression (--scores clustered) in batched launches')
    ap.add_argument('--maps', choices=['f32', 'f16'], default='f32',
                    help='feature-map / RoI-feature storage type (f32 = the metric of SURVEY 8d; f16 = BASELINE '
                         'config 5 "fp16 feature maps": float32 lerps, float16 in / out)')
    ap.add_argument('--roofline-samples', type=int, default=12, help='isolated RoI launches timed after the timed region')
    ap.add_argument('--backend', default='nccl', help="torch.distributed backend for --gpus > 1 ('nccl' = RCCL; "
                    "'gloo' only to rehearse the multi-rank path on a box with fewer GPUs than ranks)")
    ap.add_argument('--time-budget', type=float, default=190.0,
                    help='seconds the whole run aims to stay within: the accuracy gates of the detail process run on as many '
                         'scenes as fit (recorded as gate_scenes_reduced_to; --time-budget 900 runs them at full size); the '
                         'headline legs are never shortened')
    ap.add_argument('--detail-out', default=os.path.join(ROOT, 'bench_detail.json'),
                    help='side file of the N = 1 run: config 5, every end-to-end leg and every gate record (the stdout line keeps '
                         'their headline figures in `summary`)')
    ap.add_argument('--detail-child', default=None, help=argparse.SUPPRESS)    # (internal: this process IS the detail child)
    ap.add_argument('--trace', action='store_true', help='phase time stamps on stderr (synchronises at every mark)')
    ap.add_argument('--force-collective', action='store_true',
                    help='with ONE rank under torch.distributed.run: still issue the all-gather of every stream group (the RCCL '
                         'call path rehearsed on a 1-GPU box)')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # plain `python bench.py --gpus N`: this process has not touched the GPU (importing torch and counting devices do
        # not initialise it), so it starts the N ranks as fresh children and relays rank 0's line and the exit code
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit('--gpus %d but the launcher started %d rank(s) (WORLD_SIZE)' % (args.gpus, world))
    ndev = torch.cuda.device_count()
    if args.backend == 'nccl' and local_rank >= ndev:
        raise SystemExit('rank %d: local rank %d but only %d GPU(s) visible' % (rank, local_rank, ndev))
    local_dev = local_rank % max(ndev, 1)                     # (rehearsal with gloo: ranks may share a GPU)
    torch.cuda.set_device(local_dev)
    dist = None
    # (--force-collective under torchrun --nproc-per-node 1: the RCCL exchange path with one rank, a rehearsal)
    use_dist = world > 1 or (args.force_collective and 'RANK' in os.environ)
    if use_dist:
        import torch.distributed as dist
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_dev))
        else:
            dist.init_process_group(args.backend)

    from tf_eager_object_detection_amd import _lib, parallel
    from tf_eager_object_detection_amd import synthetic as syn
    from tf_eager_object_detection_amd.pipeline import FpnStreamPool, synthetic_fpn_inputs
    _lib.lib()     # fail loudly without the HIP library

    # images in flight per GPU: `streams` stream groups x `batch` images that share every kernel launch of their
    # stream (one grid dimension = image)
    S, B = max(1, args.streams), max(1, min(8, args.batch))
    R = max(1, args.rounds_per_step)
    images_per_step = R * S * B

    trace_on = args.trace

    def mark(what):
        if trace_on:
            torch.cuda.synchronize()
            print('[bench %.3f] %s' % (time.perf_counter(), what), file=sys.stderr, flush=True)

    class Workload:
        """One configuration's inputs for one score distribution (resident in HBM) + the stream pool that runs them.
        cfg: image_shape, num_classes, max_per_class, max_per_image, maps ('f32' / 'f16')."""

        def __init__(self, cfg, score_kind, nms_first_chunk, blind_chunks):
            self.cfg = cfg
            self.fdt = torch.float16 if cfg['maps'] == 'f16' else torch.float32
            fdt = self.fdt
            self.score_kind, self.nms_first_chunk, self.blind_chunks = score_kind, nms_first_chunk, blind_chunks
            self.host, dev = synthetic_fpn_inputs(cfg['image_shape'], cfg['num_classes'], NUM_PROPOSALS, CHANNELS,
                                                  seed=1234 + rank, score_kind=score_kind)
            if fdt != torch.float32:
                dev['feats'] = [f.to(fdt) for f in dev['feats']]
            # every in-flight image has its OWN inputs in HBM (slot 0 = the seeded numpy set the CPU baseline also
            # uses; the others: fresh normal feature maps, the RPN / RoI-head outputs of slot 0 row-permuted), so
            # that no image is served from lines another image pulled into L2 / Infinity Cache
            gen = torch.Generator(device='cuda')
            gen.manual_seed(4321 + rank)
            self.slot_inputs = [dev]
            for k in range(1, S * B):
                pa = torch.randperm(dev['rpn_logits'].shape[0], device='cuda', generator=gen)
                pr = torch.randperm(dev['cls_scores'].shape[0], device='cuda', generator=gen)
                self.slot_inputs.append(dict(
                    rpn_logits=dev['rpn_logits'][pa].contiguous(), rpn_deltas=dev['rpn_deltas'][pa].contiguous(),
                    feats=[torch.randn(f.shape, device='cuda', dtype=torch.float32, generator=gen).to(fdt) for f in dev['feats']],
                    cls_scores=dev['cls_scores'][pr].contiguous(), cls_deltas=dev['cls_deltas'][pr].contiguous()))
            self.pool = None
            self.exchange = None
            self.allgathers = 0
            self.plan(nms_first_chunk, blind_chunks)

        def plan(self, nms_first_chunk, blind_chunks):
            if self.pool is not None:
                self.pool.close()
            if self.exchange is not None:                       # (its communication buffers: one exchange at a time)
                self.exchange.synchronize()
                self.exchange = None
            cfg = self.cfg
            self.nms_first_chunk, self.blind_chunks = nms_first_chunk, blind_chunks
            pool = FpnStreamPool(S, cfg['image_shape'], cfg['num_classes'], NUM_PROPOSALS, CHANNELS, batch=B,
                                 blind_chunks=blind_chunks, feature_dtype=self.fdt, nms_first_chunk=nms_first_chunk,
                                 max_per_class=cfg['max_per_class'], max_per_image=cfg['max_per_image'])
            rec_len = pool.slots[0].record.numel()
            self.records = torch.zeros((pool.n, rec_len), dtype=torch.float32, device='cuda')
            for k in range(pool.n):
                pool.slots[k].record = self.records[k]            # one contiguous block per group: ONE all-gather per group
                d = self.slot_inputs[k]
                pool.bind(k, d['rpn_logits'], d['rpn_deltas'], d['feats'], d['cls_scores'], d['cls_deltas'])
            self.pool, self.rec_len = pool, rec_len
            self.exchange = parallel.GroupExchange(S, B, rec_len, 'cuda', force_collective=True) if use_dist else None
            # the inputs (generated by torch kernels on the default stream) are resident before anything is enqueued on the
            # group streams, which do not wait for the default stream by themselves
            torch.cuda.synchronize()

        def run_steps(self, nsteps):
            """nsteps steps of R rounds: in every round each of the S stream groups sends its B images through their
            shared launches; with N > 1 ranks each group's records then go through one all-gather
            (parallel.GroupExchange) -- the group's launches and its exchange are enqueued by THIS thread, in stream
            order, so the multi-rank loop has no host wait between groups either."""
            pool = self.pool
            if not use_dist:
                for _ in range(nsteps * R):
                    for g in range(S):
                        pool.submit_group(g)
                return
            gstreams = pool._group_streams
            for _ in range(nsteps * R):
                for g in range(S):
                    pool.enqueue_group(g)
                    self.exchange.gather(g, self.records[g * B:(g + 1) * B], producer_stream=gstreams[g])
                    self.allgathers += 1

        def fence(self):
            self.pool.wait()
            torch.cuda.synchronize()
            if use_dist:
                self.exchange.synchronize()
                dist.barrier()
            torch.cuda.synchronize()

        def complete(self):
            ok = 1 if all(f == 1 for f in self.pool.nms_done_all.tolist()) else 0       # (one device -> host copy)
            if use_dist:                                          # (every rank re-plans or none does)
                t = torch.tensor([ok], dtype=torch.int32, device='cuda')
                dist.all_reduce(t, op=dist.ReduceOp.MIN)
                ok = int(t.item())
            return ok == 1

        def measure(self, steps, warmup):
            """warm-up (re-planning the sync-free NMS if the score distribution needs more candidates than the plan's first
            chunk holds: first chunks of 2048 / 2560 / 3072 / 4096 candidates, then a second chunk from the ranked
            selection), then exactly `steps` timed steps between fences; max over ranks (+ every rank's own time)."""
            replans = []
            # (the ladder is fine-grained because the selection kernels' cost grows with the chunk: trained-like clustered scores
            # complete from 2560 candidates on and run 11 % faster there than with 4096; the slots' inputs are the same in the
            # warm-up and the timed region, so a plan that completes here completes there -- and is checked again after it)
            start = (self.nms_first_chunk, self.blind_chunks)
            ladder = [start] + [(n, self.blind_chunks) for n in (2048, 2560, 3072, 4096) if n > self.nms_first_chunk]
            ladder.append((4096, max(2, self.blind_chunks)))
            for first, blind in ladder:
                if (first, blind) != (self.nms_first_chunk, self.blind_chunks):
                    self.plan(first, blind)
                    replans.append({'nms_first_chunk': first, 'blind_chunks': blind})
                mark('warm-up of plan (%d, %d) on %s scores' % (first, blind, self.score_kind))
                self.run_steps(max(warmup, 1))                    # (at least one pass over every slot)
                self.fence()
                mark('warm-up done')
                if self.complete():
                    break
            else:
                raise SystemExit('NMS did not complete inside the sync-free chunks of any plan -- result would be invalid')
            self.allgathers = 0
            self.pool.nms_reruns = 0
            t0 = time.perf_counter()
            self.run_steps(steps)
            self.fence()
            # (inside the timed region) images whose sync-free NMS did not complete go through the exact mode again -- the
            # reference's NMS is always exact (model/region_proposal.py:73-81); counted in config.nms_reruns.  The plan was
            # chosen in the warm-up on the same inputs, so this finds none unless the distribution changed under it.
            self.nms_reruns = len(self.pool.recover_incomplete())
            elapsed = time.perf_counter() - t0
            mark('timed region done')
            per_rank = [elapsed]
            if use_dist:
                mine = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
                allt = torch.zeros(dist.get_world_size(), dtype=torch.float64, device='cuda')
                dist.all_gather_into_tensor(allt, mine)
                per_rank = [float(v) for v in allt.tolist()]
                elapsed = max(per_rank)
            if not self.complete():
                raise SystemExit('NMS incomplete after the exact-mode recovery -- result would be invalid')
            self.per_rank_s = per_rank
            return elapsed, replans

        def close(self):
            if self.pool is not None:
                self.pool.close()
                self.pool = None
            if self.exchange is not None:
                self.exchange.synchronize()
                self.exchange = None
            self.slot_inputs = None

    def roofline_phase(wl, samples, kernel_label, workload_key):
        """(untimed) the B-image RoI launch of stream group 0 of workload `wl`, ALONE on the GPU, cold maps.  Before every
        sample the other groups run once (their 2/3 of the inputs go through the caches; with one group a 1 GiB buffer is
        written instead), then group 0's stream waits (on the GPU) for the others.  Then the calibration kernel under the
        same protocol.  -> the `roofline` object."""
        from tf_eager_object_detection_amd import ops
        pool, cfg = wl.pool, wl.cfg
        gstreams = pool._group_streams
        flush = torch.empty(1 << 28, dtype=torch.float32, device='cuda') if S == 1 else None

        def isolate():
            if S > 1:
                for g in range(1, S):
                    pool.submit_group(g)
                pool.wait()
            else:
                with torch.cuda.stream(gstreams[0]):
                    flush.fill_(1.0)
            mine = gstreams[0]
            for st in gstreams[1:]:
                mine.wait_stream(st)
            return mine

        ev_roi = []
        for _ in range(max(3, samples)):
            isolate()
            ev = (ops.ProfEvent(), ops.ProfEvent())
            first = pool.steps[0]
            first.roi_start_event, first.roi_stop_event = ev[0].handle, ev[1].handle
            try:
                _lib.check(_lib.lib().odet_fpn_step_enqueue_batch(pool._groups[0], B, 7))    # on the group's stream
            finally:
                first.roi_start_event, first.roi_stop_event = None, None
            torch.cuda.synchronize()
            ev_roi.append(ev)
        times = [a.elapsed_ms(b) for a, b in ev_roi][2:]      # (the first two settle clocks / caches)
        roi_ms = float(np.mean(times))

        def calibrate(read_bytes, write_bytes, n=8):
            """odet_calib_stream_mix (csrc/calib.hip: a kernel that only moves these bytes with the RoI kernel's
            instructions, cache policy and image -> XCD pinning) under the SAME protocol as the RoI samples above: what
            this box's memory system needs for the launch's read : write mix."""
            rb, wb = (int(read_bytes) // 8192) * 8192, (int(write_bytes) // 8192) * 8192
            src = torch.zeros(rb // 4, dtype=torch.float32, device='cuda')
            dst = torch.empty(wb // 4, dtype=torch.float32, device='cuda')
            evs = []
            for _ in range(n):
                mine = isolate()
                ev = (ops.ProfEvent(), ops.ProfEvent())
                _lib.call('odet_calib_stream_mix', src.data_ptr(), rb, dst.data_ptr(), wb, mine.cuda_stream,
                          ev[0].handle, ev[1].handle)
                torch.cuda.synchronize()
                evs.append(ev)
            ts = [a.elapsed_ms(b) for a, b in evs][2:]
            del src, dst
            return rb, wb, float(np.mean(ts)), len(ts)

        # algorithmic bytes of the timed launch = the sum over its images (the slots of stream group 0; SURVEY 8d per image)
        elem = 2 if cfg['maps'] == 'f16' else 4
        per_slot = []
        for h_ in pool.slots[:B]:
            kk = int(h_.roi_count.item())
            per_slot.append(algorithmic_roi_bytes(h_.sorted_rois[:kk].cpu().numpy(), h_.roi_level[:kk].cpu().numpy(),
                                                  syn.fpn_level_shapes(cfg['image_shape'])[:4], cfg['image_shape'], CHANNELS,
                                                  elem=elem))
        algo = {q: sum(a_[q] for a_ in per_slot) for q in per_slot[0]}
        achieved = algo['B_roi'] / (roi_ms * 1e-3) / 1e9
        traffic, traffic_source = load_traffic(workload_key, B)
        ft = 'float' if cfg['maps'] == 'f32' else '__half'
        rf = {'bound': 'hbm', 'kernel': '%s, the %d-image launch of one stream group, alone on the GPU, cold maps' % (kernel_label, B),
              'rocprof_kernel_name': 'void k_roi_pool<1, 1, %s, 1>(RoiParams)' % ft,
              'images_per_launch': B,
              'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
              'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic, 'traffic_source': traffic_source,
              'kernel_ms': roi_ms, 'kernel_ms_samples': len(times), 'kernel_ms_min': float(np.min(times)),
              'kernel_ms_max': float(np.max(times)), 'algorithmic_bytes': algo['B_roi'],
              'B_min': algo['B_min'], 'B_taps': algo['B_taps'], 'bytes_output': algo['out']}
        # the same launch priced on the HBM bytes the PMC counters saw (reuse between RoIs served from L2 / Infinity
        # Cache is not in them; SURVEY 8d's algorithmic bytes count every RoI's cells): the plain bandwidth figure
        rf['measured_traffic_GBps'] = (traffic / (roi_ms * 1e-3) / 1e9) if (traffic and roi_ms) else None
        rf['hbm_frac_measured'] = (rf['measured_traffic_GBps'] / HBM_PEAK_GBS) if rf['measured_traffic_GBps'] else None
        rf['frac_on_B_min'] = algo['B_min'] / (roi_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
        # what the memory system of this box does with the same bytes and nothing else: B_min = every distinct map
        # cell once + the output once, moved by a kernel without arithmetic, gathers or reuse (see calibrate())
        rb, wb, cal_ms, cal_n = calibrate(algo['B_min'] - algo['out'], algo['out'])
        rf['calibration'] = {'kernel': 'k_calib_stream_mix: reads B_min - output bytes once in 1 KB rows and writes the '
                                       'output bytes (nontemporal), interleaved, XCD-pinned like the RoI launch; same '
                                       'cold protocol',
                             'bytes_read': rb, 'bytes_written': wb, 'ms': cal_ms, 'samples': cal_n,
                             'GBps': (rb + wb) / (cal_ms * 1e-3) / 1e9,
                             'roi_kernel_vs_calibration': cal_ms / roi_ms}
        return rf

    cfg3 = dict(image_shape=IMAGE_SHAPE, num_classes=NUM_CLASSES, max_per_class=50, max_per_image=50, maps=args.maps)

    def run_detail(path):
        """(the CHILD process of an N = 1 run) BASELINE configs[4], the assembled detectors end to end and their accuracy gates
        -> the side file `path`, rewritten (atomically) after every leg, so that whatever was finished when this process ends
        -- normally, by the parent's timeout, or in a GPU fault -- is on disk.  Nothing here can touch the headline: the parent
        measured it before it started this process."""
        detail = {'complete': False, 'legs_done': [], 'config5': None, 'e2e': {'conv_path': E2E_CONV_PATH}}
        e2e = detail['e2e']

        def flush(leg=None):
            if leg is not None:
                detail['legs_done'].append([leg, round(time.perf_counter() - t_main, 1)])
            with open(path + '.tmp', 'w') as f:
                json.dump(detail, f)
            os.replace(path + '.tmp', path)
            if leg is not None:
                print('[bench detail %.1f s] %s' % (time.perf_counter() - t_main, leg), file=sys.stderr, flush=True)

        def left():
            return args.time_budget - (time.perf_counter() - t_main)

        flush()
        if not args.no_config5:
            # ---- BASELINE configs[4]: 1333 x 1333, 446 118 anchors, 1000 proposals, 81 classes, caps 100 / 300 (the COCO
            # configuration's per-class / per-image caps: config/faster_rcnn_config.py:112-113), float16 feature maps
            try:
                cfg5 = dict(image_shape=(1333, 1333), num_classes=81, max_per_class=100, max_per_image=300, maps='f16')
                w5 = Workload(cfg5, 'distinct', args.nms_first_chunk, args.blind_chunks)
                steps5 = max(2, args.steps // 4)
                el5, rp5 = w5.measure(steps5, max(1, args.warmup // 2))
                rf5 = roofline_phase(w5, max(6, args.roofline_samples // 2),
                                     'k_roi_pool<MAX2, NORM_IMAGE, __half> (float16 maps: float32 lerps, float16 in / out)',
                                     'fpn_hot_path_1333x1333_r101fpn_81cls_f16maps')
                detail['config5'] = {
                    'workload': 'BASELINE configs[4]: ResNet-101-FPN hot path @ 1333x1333 (446118 anchors -> 1000 proposals), 81 '
                                'classes, max 100 per class / 300 per image, float16 feature maps into the RoI kernel',
                    'value': steps5 * images_per_step / el5, 'unit': 'img/s', 'steps': steps5, 'timed_images': steps5 * images_per_step,
                    'timed_region_s': el5, 'ms_per_image': el5 / (steps5 * images_per_step) * 1e3, 'dtype': 'f16',
                    'anchors': syn.num_fpn_anchors((1333, 1333)), 'num_classes': 81, 'max_per_class': 100, 'max_per_image': 300,
                    'nms_first_chunk': w5.nms_first_chunk, 'blind_chunks': w5.blind_chunks, 'replanned': rp5,
                    'proposals_kept': int(w5.pool.slots[0].roi_count.item()),
                    'detections_image0': int(w5.pool.slots[0].det_count.item()), 'roofline': rf5}
                w5.close()
                del w5
            except Exception as ex:
                detail['config5'] = {'error': '%s: %s' % (type(ex).__name__, ex)}
            torch.cuda.empty_cache()
            flush('config5')
        if not args.no_e2e:
            # (float16: 60 images per pass -- conv4's 50 x 84 maps then cut into four full rounds of 256-pixel workgroup tiles:
            # +2 % over 30, +5 % over 15; 8 / 16 images leave a fifth of a round empty; batch 1 -- the BASELINE configs' own
            # batch -- replayed as ONE HIP graph, batch 4 / 8 eager and as a graph; float32: 30 images per pass for the same
            # reason).  (name, dtype, batch, family, graph, eager, timed seconds)
            legs = (('fp16', 'fp16', 60, 'fpn', False, True, 3.0), ('fp16_b1', 'fp16', 1, 'fpn', True, True, 1.5),
                    ('fp32_x3', 'fp32', 30, 'fpn', False, True, 3.0), ('fp32_x3_b1', 'fp32', 1, 'fpn', True, False, 1.5),
                    ('fp32_x2', 'fp32', 30, 'fpn', False, True, 3.0), ('fp32_x2_b1', 'fp32', 1, 'fpn', True, False, 1.5),
                    ('fp32', 'fp32', 30, 'fpn', False, True, 3.0),
                    ('fp16_b4', 'fp16', 4, 'fpn', True, True, 1.0), ('fp16_b8', 'fp16', 8, 'fpn', True, True, 1.0),
                    ('fp16_resnet50_c4', 'fp16', 60, 'c4', False, True, 1.5), ('fp16_vgg16_600x800', 'fp16', 64, 'vgg16', False, True, 1.5),
                    ('fp32_x3_resnet50_c4', 'fp32', 30, 'c4', False, True, 1.5), ('fp32_x3_vgg16_600x800', 'fp32', 32, 'vgg16', False, True, 1.5))
            for name, dtn, b, fam, gr, eg, budget in legs:
                if left() < 3.0 * budget + 6.0:           # (the parent's clock: a leg that cannot finish is not started)
                    e2e[name] = {'skipped': 'time budget (%.0f s left)' % left()}
                    continue
                try:
                    e2e[name] = e2e_record(dtn, b, budget_s=budget, family=fam, graph=gr, eager=eg,
                                           f32_form='x3' if 'fp32_x3' in name else 'x2' if 'fp32_x2' in name else 'exact')
                except Exception as ex:
                    e2e[name] = {'error': '%s: %s' % (type(ex).__name__, ex)}
                flush('e2e ' + name)
            e2e['note'] = ('second record, not the headline metric: the assembled detector end to end; fp32 = parity '
                           'mode (the reference computes in float32; exact-float32 matrix instructions, 157 TFLOP/s peak, 684 GFLOP '
                           'per image), fp32_x3 / fp32_x2 = the same float32 tensors with every layer on the split-precision forms '
                           '(float32-class accuracy; three bfloat16 limbs: 417 TFLOP/s-equivalent peak, two float16 limbs: 833, '
                           'float16 range), fp16 = throughput mode, narrower than the reference, gated by map_delta_vs_fp32')
            # the accuracy side of the modes: a detector mode vs the exact-float32 detector, same weights, same annotated scenes,
            # the reference's evaluation loop (evaluation/precision_gate.py) -- for all three families.  Scenes a family needs
            # for a paired-bootstrap 95 % interval INSIDE +-0.002 (both ends; `ci_inside_bar`): ~4096 (FPN), ~6144 (C4), ~8192
            # (VGG16; the single-level detectors keep 300 proposals and fewer detections per scene); the split-precision modes
            # differ from the exact mode by float32 rounding: 512.  `--gate-images` scales all of them; the default run gives
            # the gates what is left of --time-budget (scene counts cut in proportion, recorded in gate_scenes_reduced_to), the
            # full-size gates are a longer --time-budget away (profiles/: the committed full run).
            from tf_eager_object_detection_amd.evaluation import precision_gate
            g_ = args.gate_images
            gates = [('fp16', 'fpn', g_, 'fp16', 0.0140), ('fp16_resnet50_c4', 'c4', g_ * 3 // 2, 'fp16', 0.0115),
                     ('fp16_vgg16_600x800', 'vgg16', 2 * g_, 'fp16', 0.0078),
                     ('fp32_x3', 'fpn', max(128, g_ // 8), 'x3', 0.030), ('fp32_x2', 'fpn', max(128, g_ // 8), 'x2', 0.028)]
            fixed = 2.5                                   # (two detectors built + the heads fitted, per gate)
            need = sum(n * c + fixed for _, _, n, _, c in gates)
            have = left() - 8.0
            scale = min(1.0, max(0.0, (have - fixed * len(gates)) / max(need - fixed * len(gates), 1e-9)))
            for name, fam, wanted, mode, per_scene in gates:
                n_img = wanted if scale >= 1.0 else max(128, int(wanted * scale) // 128 * 128)
                if left() < n_img * per_scene + fixed + 4.0:
                    n_img = int((left() - fixed - 4.0) / per_scene) // 128 * 128
                if n_img < 128:
                    e2e[name + '_map_delta_vs_fp32'] = {'skipped': 'time budget (%.0f s left)' % left()}
                    flush('gate %s skipped' % name)
                    continue
                try:
                    gate = precision_gate.fp16_vs_fp32(num_images=n_img, batch32=30, batch16=30, family=fam, test_mode=mode)
                    if n_img != wanted:
                        gate['gate_scenes_reduced_to'] = [n_img, wanted]
                    gate.pop('protocol', None) if name != 'fp16' else None
                    gate.pop('fit', None) if name != 'fp16' else None
                    if isinstance(e2e.get(name), dict):
                        e2e[name]['map_delta_vs_fp32'] = gate
                    else:
                        e2e[name + '_map_delta_vs_fp32'] = gate
                except Exception as ex:
                    e2e[name + '_map_delta_vs_fp32'] = {'error': '%s: %s' % (type(ex).__name__, ex)}
                flush('gate %s (%d scenes)' % (name, n_img))
        detail['complete'] = True
        flush('done')

    if args.detail_child:
        run_detail(args.detail_child)
        return

    # ---- the headline: SURVEY 8(d)'s distribution (or --scores), then the OTHER distribution beside it
    wl = Workload(cfg3, args.scores, args.nms_first_chunk, args.blind_chunks)
    elapsed, replans = wl.measure(args.steps, args.warmup)
    per_rank_s, allgathers, nms_reruns = wl.per_rank_s, wl.allgathers, wl.nms_reruns
    host = wl.host
    rf, k_kept, plan0, rec_len0 = None, int(wl.pool.slots[0].roi_count.item()), (wl.nms_first_chunk, wl.blind_chunks), wl.rec_len
    if rank == 0:
        rf = roofline_phase(wl, args.roofline_samples, 'k_roi_pool<MAX2, NORM_IMAGE> (fused crop_and_resize 14x14 + 2x2 max)',
                            'fpn_hot_path_800x1333_r101fpn_%s%s' % (args.scores, '' if args.maps == 'f32' else '_f16maps'))
        mark('roofline + calibration done')
    wl.close()                                                    # (its 2.3 GB of inputs go before the next workload is built)
    del wl
    torch.cuda.empty_cache()
    other_kind = 'clustered' if args.scores == 'distinct' else 'distinct'
    other = None
    if not args.no_second_distribution:
        wl2 = Workload(cfg3, other_kind, args.nms_first_chunk, args.blind_chunks)
        el2, rp2 = wl2.measure(args.steps, args.warmup)
        other = {'value': args.steps * images_per_step * world / el2, 'unit': 'img/s', 'ms_per_step': el2 / args.steps * 1e3,
                 'rpn_scores': other_kind, 'timed_region_s': el2, 'nms_first_chunk': wl2.nms_first_chunk,
                 'blind_chunks': wl2.blind_chunks, 'replanned': rp2, 'nms_reruns': wl2.nms_reruns,
                 'proposals_kept': int(wl2.pool.slots[0].roi_count.item())}
        wl2.close()
        del wl2
        torch.cuda.empty_cache()

    if rank == 0:
        k = k_kept
        result = {
            'metric': 'images/sec', 'value': args.steps * images_per_step * world / elapsed, 'unit': 'img/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': args.maps, 'data': 'synthetic',
            'config': {'workload': 'ResNet-101-FPN @ 800x1333 detection hot path: anchors(267069, generated in '
                                   'registers) -> fg softmax -> RegionProposal (decode+clip+exact NMS over all '
                                   'anchors, 1000 proposals) -> assign_levels -> RoI crop14x14+maxpool over '
                                   'P2..P5x256 -> post_ops (21 classes); conv backbone/heads out of scope (their '
                                   'outputs are synthetic inputs in HBM)',
                       'step': '%d rounds over the %d x %d in-flight image slots of a GPU' % (R, S, B),
                       'images_per_step_per_gpu': images_per_step, 'global_batch': images_per_step * world,
                       'timed_images': args.steps * images_per_step * world, 'timed_region_s': elapsed,
                       'rpn_scores': args.scores, 'feature_maps': args.maps, 'nms_first_chunk': plan0[0],
                       'blind_chunks': plan0[1], 'replanned': replans, 'nms_reruns': nms_reruns,
                       'streams_per_gpu': S, 'images_per_launch': B, 'images_in_flight_per_gpu': S * B,
                       'proposals_kept': k, 'parallelism': 'image-parallel x%d' % world},
            'roofline': rf,
        }
        if other is not None:
            # the same path on the other RPN score distribution, its own timed region of the same length (a trained RPN
            # clusters its best anchors: greedy NMS suppresses most of the first candidates and needs a wider chunk)
            result['value_%s' % other_kind] = other['value']
            result['ms_per_step_%s' % other_kind] = other['ms_per_step']
            result['config']['second_distribution'] = other
        # what a multi-GPU run needs to validate itself against its N ranks (weak scaling: every rank times the same number
        # of images; `value` divides by the slowest rank)
        rates = [args.steps * images_per_step / t for t in per_rank_s]
        result['multi_rank'] = {
            'world_size_env': world, 'ranks_timed': len(per_rank_s),
            'backend': (dist.get_backend() if use_dist else None),
            'rccl_world': (dist.get_world_size() if use_dist else 1),
            'collective': 'all_gather_into_tensor of the %d records of a stream group from every rank (parallel.GroupExchange)' % B
                          if use_dist else 'none (one rank: no exchange in the loop)',
            'allgathers_in_timed_region': allgathers, 'allgathers_expected': (args.steps * R * S if use_dist else 0),
            'record_bytes_per_rank_per_allgather': B * rec_len0 * 4,
            'per_rank_img_s_min': min(rates), 'per_rank_img_s_max': max(rates),
            'per_rank_spread': (max(rates) - min(rates)) / max(rates),
            'images_per_rank': args.steps * images_per_step,
        }
        summary = {'hot_path_img_s': round(result['value'], 1), 'hot_path_clustered_img_s': round(other['value'], 1) if other else None,
                   'roi_frac_B_roi': round(result['roofline']['frac'], 3),
                   'roi_frac_counter_bytes': (round(result['roofline']['hbm_frac_measured'], 3)
                                              if result['roofline']['hbm_frac_measured'] else None),
                   'roi_frac_physical': 'roi_frac_counter_bytes',
                   'roi_kernel_us': round(result['roofline']['kernel_ms'] * 1e3, 1),
                   'roi_vs_calibration': round(result['roofline']['calibration']['roi_kernel_vs_calibration'], 3)}
        if not args.no_cpu_baseline and world == 1:
            try:
                result['cpu_baseline'] = cpu_baseline(host, IMAGE_SHAPE)
                cb = result['cpu_baseline']
                summary['cpu_port_img_s'] = [round(cb['value'], 1), cb['cores']]
            except Exception as ex:                       # (the headline does not depend on the host leg)
                result['cpu_baseline'] = {'error': '%s: %s' % (type(ex).__name__, ex), 'kind': 'port', 'value': None, 'unit': 'img/s',
                                          'cores': None, 'sample': None}
            mark('cpu baseline done')
        if world == 1 and not (args.no_e2e and args.no_config5):
            # ---- everything that is not the headline runs in a FRESH CHILD PROCESS (never an exec of this one, which holds the
            # GPU) under a timeout: a hang or a GPU fault in any of its legs costs that leg, not the line.  Its records go to
            # the side file (and its progress to stderr); this line keeps their headline figures in `summary`.
            try:
                detail = run_detail_child(args, t_main)
            except Exception as ex:                       # (whatever happens around the child: the line is printed)
                detail = {'error': '%s: %s' % (type(ex).__name__, ex), 'file': None, 'complete': False, 'child_rc': None,
                          'child_s': None, 'legs_done_n': 0}
            result['detail'] = {k_: detail.get(k_) for k_ in ('file', 'complete', 'child_rc', 'child_s', 'legs_done_n')}
            try:
                summarise_detail(summary, detail)
            except Exception as ex:
                summary['detail'] = 'summary failed: %s: %s' % (type(ex).__name__, ex)
        mr = result['multi_rank']
        summary['ranks'] = [mr['rccl_world'], round(mr['per_rank_img_s_min'], 1), round(mr['per_rank_img_s_max'], 1), mr['allgathers_in_timed_region']]
        summary['run_s'] = round(time.perf_counter() - t_main, 1)
        result['summary'] = summary                   # LAST key: the tail of the line carries every headline figure
        print(compact_line(result), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

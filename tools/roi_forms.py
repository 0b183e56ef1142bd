#!/usr/bin/env python3
"""The RoI kernel in every form the reference uses it (model/roi_pooling.py), in the launch shape the pipelines issue
(the 8-image launch of a stream group: FpnStepBatch / FrcnnStepBatch, STAGE_ROI) and as the single-image launch of the
reference-surface layers, HIP events on the dispatch, cold (a 1 GiB read went through the caches) and warm.

Pricing (per launch): `B_roi` = SURVEY 8(d)'s algorithmic bytes (every RoI's unique tapped cells + its output: no credit
for reuse between RoIs -- on a 17 MB single-level map that the RoIs of an image tap 5-15 times over, B_roi / time
exceeds the HBM peak and is NOT a bandwidth), `B_min` = what HBM has to move (every tapped cell of a map once, capped by
the map, + the output) and, when profiles/<round>_roi_forms_pmc.json holds counters for the form, the HBM-side bytes
rocprofv3 saw.  Fractions are of the 8 TB/s spec peak.

    python tools/roi_forms.py [--pmc profiles/r03_roi_forms_pmc.json] > profiles/r03_roi_forms.json
    python tools/roi_forms.py --form 2 --reps 8 --cold-only          # one form (what tools/pmc_roi_forms.sh profiles)"""
import argparse, hashlib, json, os, sys
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tools._diag            # (ODET_LIB_PATH selects a diagnostic build: tools only, the product reads no environment)
from tf_eager_object_detection_amd import _lib, ops
from tf_eager_object_detection_amd import synthetic as syn
from tf_eager_object_detection_amd.pipeline import (FpnStepBatch, FrcnnStepBatch, synthetic_fpn_inputs,
                                                    synthetic_frcnn_inputs)

PEAK = 8000.0     # GB/s


def span_cells(lo, hi, dim, crop, norm, stride=None, img=None):
    lim = np.float32(dim - 1)
    if norm == 'image':
        lo_n, hi_n = np.float32(lo) / np.float32(img), np.float32(hi) / np.float32(img)
    else:
        lo_n, hi_n = (np.float32(lo) / np.float32(stride)) / lim, (np.float32(hi) / np.float32(stride)) / lim
    scale = (hi_n - lo_n) * lim / np.float32(crop - 1)
    c = lo_n * lim + np.arange(crop, dtype=np.float32) * scale
    ok = c[(c >= 0) & (c <= lim)]
    return 0 if ok.size == 0 else int(min(np.ceil(ok.max()), dim - 1) - max(np.floor(ok.min()), 0) + 1)


def bytes_single_level(rois, H, W, C, crop, elem, stride):
    """(B_roi, B_min) of one image on one H x W x C map, stride-normalised crops (roi_pooling.py:64-74)"""
    cells = sum(span_cells(r[1], r[3], H, crop, 'stride', stride) * span_cells(r[0], r[2], W, crop, 'stride', stride)
                for r in rois)
    out = len(rois) * 49 * C * elem
    return cells * C * elem + out + len(rois) * 16, min(cells, H * W) * C * elem + out + len(rois) * 16, out


def bytes_fpn(rois, levels, shapes, image, C, elem):
    """(B_roi, B_min) of one image on the pyramid, image-normalised crops (roi_pooling.py:26-35)"""
    per = {}
    tot = 0
    for r, l in zip(rois, levels):
        H, W = shapes[int(l)]
        c = span_cells(r[1], r[3], H, 14, 'image', img=image[0]) * span_cells(r[0], r[2], W, 14, 'image', img=image[1])
        tot += c
        per[int(l)] = per.get(int(l), 0) + c
    out = len(rois) * 49 * C * elem
    bmin = sum(min(c, shapes[l][0] * shapes[l][1]) for l, c in per.items()) * C * elem + out + len(rois) * 16
    return tot * C * elem + out + len(rois) * 16, bmin, out


FORMS = [
    # name, kind, image, channels, proposals, max_pooling_flag / pool mode, map dtype
    ('config 3: ResNet-101-FPN, 14x14 + 2x2 max over P2..P5 x 256, float32 maps, 8 x 1000 RoIs (the bench kernel)', 'fpn', (800, 1333), 256, 1000, None, 'f32'),
    ('config 5 arrangement: the same with float16 maps / features, 8 x 1000 RoIs', 'fpn', (800, 1333), 256, 1000, None, 'f16'),
    ('config 2: ResNet-50 C4, RoiPoolingCropAndResize crop 7x7 (no pooling), 50x84x1024, 8 x 300 RoIs', 'frcnn', (800, 1333), 1024, 300, False, 'f32'),
    ('config 2 variant: 14x14 + 2x2 max on the C4 map, 50x84x1024, 8 x 300 RoIs', 'frcnn', (800, 1333), 1024, 300, True, 'f32'),
    ('config 1: VGG16, 14x14 + 2x2 max, 38x50x512 (600x800), 8 x 300 RoIs', 'frcnn', (600, 800), 512, 300, True, 'f32'),
    ('config 2 with float16 maps (what the float16 C4 detector feeds), crop 7x7, 8 x 300 RoIs', 'frcnn', (800, 1333), 1024, 300, False, 'f16'),
    ('tensorpack RoIAlign (TP_ALIGN, 2x2 avg), 50x84x1024, ONE image x 300 RoIs (reference-surface layer)', 'tp', (800, 1333), 1024, 300, None, 'f32'),
]


def build(form, B):
    name, kind, image, C, K, flag, mdt = form
    fdt = torch.float16 if mdt == 'f16' else torch.float32
    elem = 2 if mdt == 'f16' else 4
    gen = torch.Generator(device='cuda'); gen.manual_seed(99)
    if kind == 'fpn':
        sb = FpnStepBatch(B, image, 21, K, C, feature_dtype=fdt, blind_chunks=1)
        host, dev = synthetic_fpn_inputs(image, 21, K, C, seed=1234)
        keep = []
        for b in range(B):
            pa = torch.randperm(dev['rpn_logits'].shape[0], device='cuda', generator=gen)
            feats = [torch.randn(f.shape, device='cuda', dtype=torch.float32, generator=gen).to(fdt) for f in dev['feats']]
            d = (dev['rpn_logits'][pa].contiguous(), dev['rpn_deltas'][pa].contiguous(), feats, dev['cls_scores'], dev['cls_deltas'])
            keep.append(d)
            sb.bind(b, *d)
        sb.enqueue(sb.STAGE_PROPOSALS, B)
        torch.cuda.synchronize()
        shapes = syn.fpn_level_shapes(image)[:4]
        algo = [bytes_fpn(h.sorted_rois[:int(h.roi_count.item())].cpu().numpy(), h.roi_level[:int(h.roi_count.item())].cpu().numpy(),
                          shapes, image, C, elem) for h in sb.slots]
        map_bytes = B * sum(h * w for h, w in shapes) * C * elem
    elif kind == 'frcnn':
        sb = FrcnnStepBatch(B, image, 21, K, C, max_pooling_flag=flag, feature_dtype=fdt, blind_chunks=2)
        keep = []
        for b in range(B):
            host, dev = synthetic_frcnn_inputs(image, 21, K, C, seed=500 + b)
            d = (dev['rpn_logits'], dev['rpn_deltas'], dev['feat'].to(fdt), dev['cls_scores'], dev['cls_deltas'])
            keep.append(d)
            sb.bind(b, *d)
        sb.enqueue(sb.STAGE_PROPOSALS, B)
        torch.cuda.synchronize()
        h0 = sb.slots[0]
        crop = 14 if flag else 7
        algo = [bytes_single_level(h.rois[:int(h.roi_count.item())].cpu().numpy(), h.fh, h.fw, C, crop, elem, 16.0) for h in sb.slots]
        map_bytes = B * h0.fh * h0.fw * C * elem
    else:
        sb = None
        rng = np.random.default_rng(7)
        hgt = np.exp(rng.uniform(np.log(32), np.log(600), K)); wid = hgt * np.exp(rng.uniform(-0.7, 0.7, K))
        cy, cx = rng.uniform(0, image[0], K), rng.uniform(0, image[1], K)
        bx = np.stack([cx - wid / 2, cy - hgt / 2, cx + wid / 2, cy + hgt / 2], 1)
        bx[:, 0::2] = np.clip(bx[:, 0::2], 0, image[1] - 1); bx[:, 1::2] = np.clip(bx[:, 1::2], 0, image[0] - 1)
        bx = bx.astype(np.float32)
        maps = [torch.randn((1, 50, 84, C), device='cuda', generator=gen)]
        rois = torch.from_numpy(bx).cuda()
        level = torch.zeros(K, dtype=torch.int32, device='cuda')
        order = ops.roi_order(rois, level, image)
        out = torch.empty((K, 7, 7, C), dtype=torch.float32, device='cuda')
        keep = (maps, rois, level, order, out)
        algo = [bytes_single_level(bx, 50, 84, C, 14, 4, 16.0)]
        map_bytes = 50 * 84 * C * 4
        B = 1
    return sb, keep, algo, map_bytes, B


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--form', type=int, default=-1)
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--reps', type=int, default=14)
    ap.add_argument('--cold-only', action='store_true')
    ap.add_argument('--pmc', default=os.path.join(ROOT, 'profiles', 'r04_roi_forms_pmc.json'))
    a = ap.parse_args()
    _lib.lib()
    flush = torch.ones(1 << 28, dtype=torch.float32, device='cuda')
    sink = torch.zeros(4, dtype=torch.int32, device='cuda')
    pmc = {}
    if os.path.exists(a.pmc):
        pmc = json.load(open(a.pmc)).get('forms', {})
    res = {'lib': os.environ.get('ODET_LIB_PATH', 'in-tree'), 'peak_GBps': PEAK,
           'what': 'k_roi_pool per form and launch shape: us cold (after a 1 GiB read) / warm; B_roi (SURVEY 8d, reuse-blind), '
                   'B_min (what HBM has to move), counter bytes (rocprofv3 --pmc, FETCH_SIZE x measured factor + WRITE_SIZE); '
                   'fractions of 8 TB/s'}
    for i, form in enumerate(FORMS):
        if a.form >= 0 and i != a.form:
            continue
        sb, keep, algo, map_bytes, B = build(form, a.batch)
        if a.form >= 0:
            # counter calibration in the same profiler pass: known byte counts at both lane widths
            for bpl in (16, 8):
                _lib.call('odet_calib_read_rows', flush.data_ptr(), 1 << 29, bpl, sink.data_ptr(), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()

        def launch(ev):
            if sb is not None:
                sb.steps[0].roi_start_event, sb.steps[0].roi_stop_event = ev[0].handle, ev[1].handle
                sb.enqueue(sb.STAGE_ROI, B)
                sb.steps[0].roi_start_event, sb.steps[0].roi_stop_event = None, None
            else:
                maps, rois, level, order, out = keep
                ops.roi_pool(maps, rois, level, ops.ROI_NORM_TP_ALIGN, 7, ops.ROI_POOL_AVG2, strides=[16.0],
                             image_shape=form[2], out=out, events=ev, order=order)
        times = {}
        for mode in (('cold',) if a.cold_only else ('cold', 'warm')):
            ts = []
            for _ in range(a.reps):
                if mode == 'cold':
                    flush.sum()
                ev = (ops.ProfEvent(), ops.ProfEvent())
                launch(ev)
                torch.cuda.synchronize()
                ts.append(ev[0].elapsed_ms(ev[1]) * 1e3)
            times[mode] = float(np.median(ts[2:]))
        b_roi = sum(x[0] for x in algo); b_min = sum(x[1] for x in algo); out_b = sum(x[2] for x in algo)
        cold = times['cold']
        rec = dict(images_per_launch=B, rois_per_launch=int(sum((x[2] // (49 * form[3] * (2 if form[6] == 'f16' else 4))) for x in algo)),
                   us_cold=round(cold, 1), us_warm=round(times.get('warm', 0.0), 1) if 'warm' in times else None,
                   B_roi_MB=round(b_roi / 1e6, 1), B_min_MB=round(b_min / 1e6, 1), map_MB=round(map_bytes / 1e6, 1),
                   out_MB=round(out_b / 1e6, 1),
                   GBps_cold_on_B_roi=round(b_roi / cold / 1e3), GBps_cold_on_B_min=round(b_min / cold / 1e3),
                   frac_of_peak_on_B_min=round(b_min / cold / 1e3 / PEAK, 3))
        c = pmc.get(str(i))
        if c:
            rec['counter_MB'] = round(c['hbm_bytes_per_launch'] / 1e6, 1)
            rec['counter_read_MB'] = round(c['read_bytes_per_launch'] / 1e6, 1)
            rec['counter_write_MB'] = round(c['write_bytes_per_launch'] / 1e6, 1)
            rec['frac_of_peak_on_counter_bytes'] = round(c['hbm_bytes_per_launch'] / cold / 1e3 / PEAK, 3)
            rec['counter_over_B_min'] = round(c['hbm_bytes_per_launch'] / b_min, 3)
        if sb is not None:
            hsh = hashlib.sha256()
            for b in range(B):
                hsh.update(sb.roi_features[b].cpu().numpy().tobytes())
            rec['sha256'] = hsh.hexdigest()[:16]
        res['%d: %s' % (i, form[0])] = rec
        del sb, keep
        torch.cuda.empty_cache()
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
    main()

#!/usr/bin/env python3
"""The RoI kernel in every form the reference uses it (model/roi_pooling.py), at the configs' shapes: one launch per
form, HIP events on the dispatch, cold (a 1 GiB read went through the caches) and warm, next to the form's algorithmic
bytes (the RoI's unique tapped cells + its output) and to the previous build of the library when
ODET_LIB_PATH points at one.  RoIs: uniform random boxes with the size distribution of RPN proposals of that config
(log-uniform edge lengths), spatially ordered as the path orders them.  Prints one JSON object.

    python tools/roi_forms.py > profiles/r02_roi_forms.json"""
import hashlib, json, os, sys
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tf_eager_object_detection_amd import _lib, ops


def unique_cells(rois, H, W, crop, norm, stride=None, image=None):
    """unique cells tapped by each RoI on an H x W map (first / last in-bounds sample of the crop grid; TF
    crop_and_resize with the reference's normalisation: roi_pooling.py:26-35 (image) / 64-74 (stride))"""
    tot = 0
    for r in rois:
        spans = []
        for lo, hi, dim, img in ((r[1], r[3], H, image[0] if image else 0), (r[0], r[2], W, image[1] if image else 0)):
            lim = np.float32(dim - 1)
            if norm == 'image':
                lo_n, hi_n = np.float32(lo) / np.float32(img), np.float32(hi) / np.float32(img)
            else:
                lo_n, hi_n = (np.float32(lo) / np.float32(stride)) / lim, (np.float32(hi) / np.float32(stride)) / lim
            scale = (hi_n - lo_n) * lim / np.float32(crop - 1)
            c = lo_n * lim + np.arange(crop, dtype=np.float32) * scale
            ok = c[(c >= 0) & (c <= lim)]
            spans.append(0 if ok.size == 0 else int(min(np.ceil(ok.max()), dim - 1) - max(np.floor(ok.min()), 0) + 1))
        tot += spans[0] * spans[1]
    return tot


def boxes(rng, n, image, lo_edge, hi_edge):
    h = np.exp(rng.uniform(np.log(lo_edge), np.log(hi_edge), n)); w = h * np.exp(rng.uniform(-0.7, 0.7, n))
    cy, cx = rng.uniform(0, image[0], n), rng.uniform(0, image[1], n)
    b = np.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], 1)
    b[:, 0::2] = np.clip(b[:, 0::2], 0, image[1] - 1); b[:, 1::2] = np.clip(b[:, 1::2], 0, image[0] - 1)
    return b.astype(np.float32)


def timed(fn, flush, reps=14):
    cold, warm = [], []
    for mode, acc in (('cold', cold), ('warm', warm)):
        for _ in range(reps):
            if mode == 'cold':
                flush.sum()
            a, b = ops.ProfEvent(), ops.ProfEvent()
            fn((a, b))
            torch.cuda.synchronize()
            acc.append(a.elapsed_ms(b) * 1e3)
    return float(np.median(cold[2:])), float(np.median(warm[2:]))


def main():
    _lib.lib()
    rng = np.random.default_rng(7)
    g = torch.Generator(device='cuda'); g.manual_seed(7)
    flush = torch.ones(1 << 28, dtype=torch.float32, device='cuda')
    res = {'lib': os.environ.get('ODET_LIB_PATH', 'in-tree'),
           'what': 'k_roi_pool per form: us cold (after a 1 GiB read) / warm, algorithmic bytes B_roi = unique cells x C x 4 '
                   '+ output, GB/s on them, SHA-256 of the output'}
    forms = [
        ('config 2: ResNet-50 C4, RoiPoolingCropAndResize 14x14 + 2x2 max, 50x84x1024, 300 RoIs', (800, 1333), [(50, 84)], 1024, 300, 'stride', ops.ROI_POOL_MAX2, 16.0, (32, 600)),
        ('config 2 un-pooled: crop 7x7 (POOL_NONE), 50x84x1024, 300 RoIs', (800, 1333), [(50, 84)], 1024, 300, 'stride', ops.ROI_POOL_NONE, 16.0, (32, 600)),
        ('config 1: VGG16, 14x14 + 2x2 max, 38x50x512 (600x800), 300 RoIs', (600, 800), [(38, 50)], 512, 300, 'stride', ops.ROI_POOL_MAX2, 16.0, (32, 500)),
        ('tensorpack RoIAlign (TP_ALIGN, 2x2 avg), 50x84x1024, 300 RoIs', (800, 1333), [(50, 84)], 1024, 300, 'tp', ops.ROI_POOL_AVG2, 16.0, (32, 600)),
        ('config 2 at training size: 2000 RoIs, 50x84x1024, 14x14 + 2x2 max', (800, 1333), [(50, 84)], 1024, 2000, 'stride', ops.ROI_POOL_MAX2, 16.0, (32, 600)),
    ]
    for name, image, shapes, C, n, norm, pool, stride, edges in forms:
        maps = [torch.randn((1, h, w, C), device='cuda', generator=g) for h, w in shapes]
        b = boxes(rng, n, image, *edges)
        rois = torch.from_numpy(b).cuda()
        level = torch.zeros(n, dtype=torch.int32, device='cuda')
        order = ops.roi_order(rois, level, image)
        nm = {'stride': ops.ROI_NORM_STRIDE, 'tp': ops.ROI_NORM_TP_ALIGN}[norm]
        out = torch.empty((n, 7, 7, C), dtype=torch.float32, device='cuda')
        def run(ev):
            ops.roi_pool(maps, rois, level, nm, 7, pool, strides=[stride], image_shape=image, out=out, events=ev, order=order)
        cold, warm = timed(run, flush)
        crop = 7 if pool == ops.ROI_POOL_NONE else 14
        uc = unique_cells(b, shapes[0][0], shapes[0][1], crop, 'stride', stride=stride)
        b_roi = uc * C * 4 + n * 49 * C * 4 + n * 16
        res[name] = dict(us_cold=round(cold, 1), us_warm=round(warm, 1), B_roi_MB=round(b_roi / 1e6, 1),
                         map_MB=round(shapes[0][0] * shapes[0][1] * C * 4 / 1e6, 1), out_MB=round(n * 49 * C * 4 / 1e6, 1),
                         GBps_cold_on_B_roi=round(b_roi / cold / 1e3), GBps_warm_on_B_roi=round(b_roi / warm / 1e3),
                         sha256=hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16])
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
    main()

#!/usr/bin/env python3
"""CPU only (oracle): the cell footprints of the single-level RoI workloads (BASELINE configs 1-2: VGG16 600 x 800, ResNet C4
800 x 1333; 300 proposals of the exact RPN NMS on the SURVEY 8(d) synthetic inputs) -- what share of the RoIs an LDS-staged form
(footprint <= 12 x 12 cells of a 128-channel slice = 73 KB) could take.  VERDICT r4 next #6; DESIGN section 3.2.
Round 5: VGG16 25.8 % fit 12 x 12 (49.5 % fit 16 x 16), median footprint 195 cells; C4 17.1 % (41.2 %), median 285 cells."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import c_oracle as co
from tf_eager_object_detection_amd import synthetic as syn
from tf_eager_object_detection_amd.utils.anchor_generator import generate_anchor_base

for name, shape, K in (('vgg16 600x800', (600, 800), 300), ('c4 800x1333', (800, 1333), 300)):
    tot = fit12 = fit16 = 0
    cells, spacing = [], []
    for seed in range(4):
        rng = np.random.default_rng(1234 + seed)
        fh, fw = int(math.ceil(shape[0] / 16)), int(math.ceil(shape[1] / 16))
        n = fh * fw * 9
        logits = rng.normal(0, 2.0, (fh * fw, 18)).astype(np.float32)
        deltas = syn.rpn_deltas(n, rng, 0.1)
        anchors = co.anchors_shift(generate_anchor_base(16, (0.5, 1, 2), (8, 16, 32)).astype(np.float32), 16, fh, fw)
        rois, _ = co.region_proposal(deltas, anchors, co.rpn_fg_frcnn(logits, 9), shape, K, 0.7)
        w, h = (rois[:, 2] - rois[:, 0]) / 16.0, (rois[:, 3] - rois[:, 1]) / 16.0
        cw, ch = np.minimum(np.ceil(w) + 2, fw), np.minimum(np.ceil(h) + 2, fh)
        tot += len(rois)
        fit12 += int(np.sum((cw <= 12) & (ch <= 12)))
        fit16 += int(np.sum((cw <= 16) & (ch <= 16)))
        cells += list(cw * ch)
        spacing += list(np.sqrt(w * h) / 13.0)               # cells between neighbouring samples of the 14 x 14 grid
    sp = np.asarray(spacing)
    print('%-14s rois %d | footprint fits 12x12 cells: %.1f %% | 16x16: %.1f %% | median %d cells, mean %d | sample spacing >= 1 cell '
          '(no tap shared between samples): %.1f %%' % (name, tot, 100.0 * fit12 / tot, 100.0 * fit16 / tot, np.median(cells), np.mean(cells),
                                                       100.0 * np.mean(sp >= 1.0)))

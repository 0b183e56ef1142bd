#!/usr/bin/env python3
"""End-to-end ResNet-101-FPN inference (dense parts through PyTorch-ROCm library convolutions, everything
between them through the HIP hot path) on synthetic 800x1333 images, random-init weights.
SURVEY.md 8(f) ranks 2-3; not the bench.py metric (that one is the hot path alone).

    python tools/e2e_bench.py [--dtype fp16|fp32] [--batch B] [--steps K] [--depth 101]"""
import argparse, json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tf_eager_object_detection_amd.model.fpn_detector import ResNetFpnDetector
from tf_eager_object_detection_amd.model.frcnn_detector import ResNetC4Detector, Vgg16Detector

ap = argparse.ArgumentParser()
ap.add_argument('--dtype', default='fp16')
ap.add_argument('--batch', type=int, default=1)
ap.add_argument('--steps', type=int, default=30)
ap.add_argument('--warmup', type=int, default=5)
ap.add_argument('--depth', type=int, default=101)
ap.add_argument('--model', default='fpn', choices=['fpn', 'c4', 'vgg16'],
                help='fpn: ResNet-FPN (1000 proposals); c4: ResNet-C4 Faster R-CNN (300 proposals); vgg16: VGG16 Faster R-CNN')
ap.add_argument('--h', type=int, default=800)
ap.add_argument('--w', type=int, default=1333)
ap.add_argument('--miopen-find', action='store_true', help='torch.backends.cudnn.benchmark = True (no effect since round 4: the detectors launch no library convolution)')
ap.add_argument('--f32-form', default='exact', choices=['exact', 'x3', 'x2'], help='float32 mode: exact-float32 matrix instructions or the '
                'split-precision form (csrc/conv_x3.hip)')
ap.add_argument('--graph', action='store_true', help='replay the whole forward pass as one HIP graph')
ap.add_argument('--blind-chunks', type=int, default=3, help='FPN: sync-free NMS chunks (3: the third one on the full order)')
ap.add_argument('--per-image', action='store_true', help='FPN: every image through its own hot-path launches and RoI-head call '
                'instead of the batched launches (FpnStepBatch)')
a = ap.parse_args()
dt = {'fp16': torch.float16, 'bf16': torch.bfloat16, 'fp32': torch.float32}[a.dtype]
torch.backends.cudnn.benchmark = bool(a.miopen_find)
torch.manual_seed(0)
if a.model == 'fpn':
    # chunk 0 + chunk 1 from the ranked selection + one per-image chunk on the full order: the float16 logits of the
    # random-init RPN tie in thousands, which the selection cannot split
    hot_kw = dict(blind_chunks=a.blind_chunks, batched=not a.per_image)
    model = ResNetFpnDetector(a.depth, 21, (a.h, a.w), 1000, dtype=dt, max_batch=a.batch, f32_form=a.f32_form, **hot_kw).prepare()
elif a.model == 'c4':
    model = ResNetC4Detector(a.depth, 21, (a.h, a.w), 300, dtype=dt, max_batch=a.batch, blind_chunks=4, f32_form=a.f32_form).prepare()
else:
    model = Vgg16Detector(21, (a.h, a.w), 300, dtype=dt, max_batch=a.batch, blind_chunks=4, f32_form=a.f32_form).prepare()
rng = np.random.default_rng(0)
img = (rng.uniform(0, 255, (a.batch, a.h, a.w, 3)) - np.float32([103.939, 116.779, 123.68])).astype(np.float32)
img = torch.from_numpy(img).cuda()
t0 = time.perf_counter()
for _ in range(a.warmup):
    out = model(img)
torch.cuda.synchronize()
step = model.capture(a.batch) if a.graph else model
if a.graph:
    for _ in range(3):
        out = step(img)
    torch.cuda.synchronize()
t_warm = time.perf_counter() - t0
t0 = time.perf_counter()
for _ in range(a.steps):
    out = step(img)
torch.cuda.synchronize()
el = time.perf_counter() - t0
# per-part timing (one extra pass each, synchronised)
def timed(fn, n=5):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): r = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3, r
with torch.no_grad():
    t_feat, p = timed(lambda: model.features(img))
    t_rpn, _ = timed(lambda: model.rpn(p))
if not isinstance(p, (tuple, list)):
    p = [p]
print(json.dumps({'metric': 'end-to-end images/sec', 'value': a.steps * a.batch / el, 'unit': 'img/s',
                  'model': {'fpn': 'ResNet-%d-FPN' % a.depth, 'c4': 'ResNet-%d-C4 Faster R-CNN' % a.depth, 'vgg16': 'VGG16 Faster R-CNN'}[a.model], 'image': [a.h, a.w], 'dtype': a.dtype, 'batch': a.batch,
                  'ms_per_image': el / (a.steps * a.batch) * 1e3, 'warmup_s': t_warm,
                  'ms_backbone_neck_per_batch': t_feat, 'ms_rpn_head_per_batch': t_rpn,
                  'detections_image0': int(out[0][3].item()), 'finite': bool(torch.isfinite(p[0]).all().item()), 'miopen_find': bool(a.miopen_find), 'hip_graph': bool(a.graph), 'hot_path': 'per-image' if a.per_image else 'batched',
                  'nms_done': [int(h.nms_done.item()) for h in model._hot]}))

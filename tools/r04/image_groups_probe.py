#!/usr/bin/env python3
"""Round 4: do the HBM-bound first stages (stem, conv2, conv3 of ResNet-101, float16, 800x1333) run faster when a large
batch goes through them in groups of g images, so that a group's maps (34 MB per image for conv2's 256-channel map) stay in
the 256 MB Infinity Cache from the launch that writes them to the launch that reads them?

    python tools/r04/image_groups_probe.py [batch, default 30]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tf_eager_object_detection_amd.model import fpn_detector as fpn

B = int(sys.argv[1]) if len(sys.argv) > 1 else 30
torch.manual_seed(0)
det = fpn.ResNetFpnDetector(depth=101, image_shape=(800, 1333), dtype=torch.float16).cuda().eval().prepare()
images = torch.rand(B, 800, 1333, 3, device='cuda') * 255 - 120


def timed(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / n


res = {}
with torch.no_grad():
    x_all = fpn._stem(det.conv1, images, det.dtype)
    c2_all = det.conv2(x_all)
    c3_all = det.conv3(c2_all)
    for g in (1, 2, 3, 4, 5, 6, 10, 15, 30):
        if g > B:
            continue
        groups = [(i, min(B, i + g)) for i in range(0, B, g)]

        def stem_conv2():
            return [det.conv2(fpn._stem(det.conv1, images[a:b], det.dtype)) for a, b in groups]

        def conv2_only():
            return [det.conv2(x_all[a:b]) for a, b in groups]

        def conv3_only():
            return [det.conv3(c2_all[a:b]) for a, b in groups]

        def to_conv3():
            return [det.conv3(det.conv2(fpn._stem(det.conv1, images[a:b], det.dtype))) for a, b in groups]

        def conv4_only():
            return [det.conv4(c3_all[a:b]) for a, b in groups]

        same = all(torch.equal(y, c2_all[a:b]) for y, (a, b) in zip(conv2_only(), groups))
        row = {'stem+conv2': round(timed(stem_conv2), 3), 'conv2': round(timed(conv2_only), 3),
               'conv3': round(timed(conv3_only), 3), 'stem+conv2+conv3': round(timed(to_conv3), 3), 'conv4': round(timed(conv4_only), 3), 'identical': same}
        res['groups of %d' % g] = row
        print(g, json.dumps(row), flush=True)
os.makedirs('gpurun_out', exist_ok=True)
json.dump({'batch': B, 'ms': res}, open('gpurun_out/r04_image_groups.json', 'w'), indent=1)

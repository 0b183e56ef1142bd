#!/usr/bin/env python3
"""Accuracy gate of the float16 throughput mode (evaluation/precision_gate.py): float16 vs float32 ResNet-101-FPN
detector on identical seeded weights and synthetic images, scored with the reference's evaluation loop.

    python tools/fp16_accuracy.py [--images 32] [--depth 101] [--h 800 --w 1333] [--seed 0]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD', '0')
import torch
from tf_eager_object_detection_amd.evaluation import precision_gate as pg

ap = argparse.ArgumentParser()
ap.add_argument("--images", type=int, default=256)
ap.add_argument("--train-images", type=int, default=64)
ap.add_argument("--ridge", type=float, nargs="+", default=[1e-3])
ap.add_argument('--depth', type=int, default=101)
ap.add_argument('--h', type=int, default=800)
ap.add_argument('--w', type=int, default=1333)
ap.add_argument('--seed', type=int, default=0)
ap.add_argument('--proposals', type=int, default=1000)
ap.add_argument('--miopen-find', action='store_true')
a = ap.parse_args()
torch.backends.cudnn.benchmark = bool(a.miopen_find)
for r in a.ridge:
    rec = pg.fp16_vs_fp32(a.images, (a.h, a.w), a.depth, num_proposals=a.proposals, seed=a.seed, train_images=a.train_images, ridge=r)
    rec.pop('protocol'); rec.pop('weights')
    print(json.dumps(rec), flush=True)

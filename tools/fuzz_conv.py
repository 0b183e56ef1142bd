#!/usr/bin/env python3
"""Seeded sweep of the hand-written convolutions against torch on integer-valued data (exact comparison): random map
sizes (incl. one-pixel rows / columns), batches, channel counts, so that every workgroup-tile choice of conv3x3_launch
and the edge handling of the fused forms (RpnHead, bottleneck tail with 64 / 128 / 256 middle channels, stem, VGG16's first
convolution, convolution + pooling) and of the pointwise forms (1x1 / strided / shortcut, lateral + merge, two sources along K;
float16 and float32) are exercised.

    python tools/fuzz_conv.py [--cases N] [--seed S]"""
import argparse, os, sys
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tf_eager_object_detection_amd import ops


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cases', type=int, default=210)
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--f32-form', default='exact', choices=['exact', 'x3', 'x2'], help='the form the float32 cases run on (x3: the '
                    'split-precision kernel is exact on integer data too)')
    a = ap.parse_args()
    ops.f32_form(a.f32_form).__enter__()          # (for the rest of the process)
    g = torch.Generator(device='cuda'); g.manual_seed(a.seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), device='cuda', generator=g).item())
    sparse = lambda shape, pct, lo, hi: ((torch.randint(0, 100, shape, device='cuda', generator=g) < pct).half()
                                         * torch.randint(lo, hi + 1, shape, device='cuda', generator=g).half())
    for case in range(a.cases):
        kind = case % 9
        B, H, W = ri(1, 3), ri(1, 70), ri(1, 90)
        if kind == 7:      # VGG16's first convolution straight from the image (round 3)
            img = torch.randint(-3, 4, (B, H, W, 3), device='cuda', generator=g).to(torch.float16 if case & 8 else torch.float32)
            w = torch.randint(-2, 3, (64, 3, 3, 3), device='cuda', generator=g).half()
            b = torch.randint(-3, 4, (64,), device='cuda', generator=g).half()
            relu = bool(case & 16)
            got = ops.conv3x3_rgb(img, ops.conv3x3_rgb_pack_weights(w), b, relu=relu)
            want = F.conv2d(img.permute(0, 3, 1, 2).float(), w.float(), b.float(), 1, 1)
            want = (F.relu(want) if relu else want).permute(0, 2, 3, 1)
            assert torch.equal(got.float(), want), ('conv3x3_rgb', B, H, W)
            continue
        if kind == 8:      # 3x3 convolution + ReLU + 2x2 'same' max-pooling in one launch (round 3)
            cin, cout = 64 * ri(1, 4), 64 * ri(1, 8)
            x = torch.randint(-2, 3, (B, H, W, cin), device='cuda', generator=g).half()
            w = sparse((cout, cin, 3, 3), 8, -2, 2).contiguous(memory_format=torch.channels_last)
            b = torch.randint(-3, 4, (cout,), device='cuda', generator=g).half()
            got = ops.conv3x3_relu_pool2_f16(x, w, b)
            want = F.relu(F.conv2d(x.permute(0, 3, 1, 2).float(), w.float(), b.float(), 1, 1))
            want = F.max_pool2d(want, 2, 2, ceil_mode=True).permute(0, 2, 3, 1)
            assert float(want.abs().max()) < 2048 and torch.equal(got.float(), want), ('conv3x3_relu_pool2', B, H, W, cin, cout)
            continue
        if 4 <= kind <= 6:      # pointwise forms (round 3), float16 and float32: 1x1 / strided / shortcut, lateral + merge, two sources
            f32 = bool(case & 8)
            dt = torch.float32 if f32 else torch.float16
            gran = 32 if f32 else 64
            stride = ri(1, 2)
            K, N = gran * ri(2, 12), 64 * ri(1, 9)
            Ho, Wo = (H + stride - 1) // stride, (W + stride - 1) // stride
            x = sparse((B, H, W, K), 30, -2, 2).to(dt)
            w = sparse((N, K), 40, -2, 2).to(dt)
            b = torch.randint(-3, 4, (N,), device='cuda', generator=g).to(dt)
            xs = x[:, ::stride, ::stride].double()
            if kind == 4:
                r = torch.randint(-4, 5, (B, Ho, Wo, N), device='cuda', generator=g).to(dt) if case & 16 else None
                want = xs @ w.double().t() + b.double() + (r.double() if r is not None else 0)
                want = torch.relu(want) if case & 32 else want
                got = ops.pointwise(x, w, b, r, bool(case & 32), stride)
                what = ('pointwise', dt, B, H, W, K, N, stride)
            elif kind == 5:
                h2, w2 = max(1, (H + 1) // 2), max(1, (W + 1) // 2)
                top = torch.randint(-8, 9, (B, h2, w2, N), device='cuda', generator=g).to(dt)
                lat = (x.double() @ w.double().t() + b.double()).to(dt)
                want = ops.fpn_topdown_merge(top, lat).double()
                got = ops.lateral_merge(x, w, b, top)
                what = ('lateral_merge', dt, B, H, W, K, N)
            else:
                K1 = gran * ri(1, 6)
                a1 = sparse((B, Ho, Wo, K1), 30, -2, 2).to(dt)
                w1 = sparse((N, K1), 40, -2, 2).to(dt)
                want = torch.relu(a1.double() @ w1.double().t() + xs @ w.double().t() + b.double())
                got = ops.pointwise_dual(a1, x, torch.cat([w1, w], 1).contiguous(), b, stride, relu=True)
                what = ('pointwise_dual', dt, B, H, W, K1, K, N, stride)
            assert float(want.abs().max()) < 2048 and torch.equal(got.double(), want), what
            continue
        if kind == 0 and case & 8:      # float32 3x3, any cout % 64
            cin, cout = 32 * ri(1, 6), 64 * ri(1, 6)
            x = torch.randint(-2, 3, (B, H, W, cin), device='cuda', generator=g).float()
            w = sparse((cout, cin, 3, 3), 8, -2, 2).float().contiguous(memory_format=torch.channels_last)
            b = torch.randint(-3, 4, (cout,), device='cuda', generator=g).float()
            got = ops.conv3x3_f32(x, w, b, relu=bool(case & 4))
            want = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), b.double(), 1, 1)
            want = (F.relu(want) if case & 4 else want).permute(0, 2, 3, 1)
            assert torch.equal(got.double(), want), ('conv3x3_f32', B, H, W, cin, cout)
            continue
        if kind == 0:      # plain 3x3, any cout % 64
            cin, cout = 64 * ri(1, 4), 64 * ri(1, 10)
            x = torch.randint(-2, 3, (B, H, W, cin), device='cuda', generator=g).half()
            w = sparse((cout, cin, 3, 3), 8, -2, 2).contiguous(memory_format=torch.channels_last)
            b = torch.randint(-3, 4, (cout,), device='cuda', generator=g).half()
            got = ops.conv3x3_f16(x, w, b, relu=bool(case & 4))
            want = F.conv2d(x.permute(0, 3, 1, 2).float(), w.float(), b.float(), 1, 1)
            want = (F.relu(want) if case & 4 else want).permute(0, 2, 3, 1)
            assert float(want.abs().max()) < 2048 and torch.equal(got.float(), want), ('conv3x3', B, H, W, cin, cout)
        elif kind == 1:    # fused RpnHead over 1..4 maps
            A, cin, cout = ri(1, 5), 64 * ri(1, 3), 256 * ri(1, 2)
            shapes = [(max(1, H >> l), max(1, W >> l)) for l in range(ri(1, 4))]
            xs = [sparse((B, h, w, cin), 4, 1, 1) for h, w in shapes]
            w3 = sparse((cout, cin, 3, 3), 4, -2, 2).contiguous(memory_format=torch.channels_last)
            b3 = torch.randint(-3, 4, (cout,), device='cuda', generator=g).half()
            w1 = sparse((6 * A, cout), 10, -2, 2)
            b1 = torch.randint(-3, 4, (6 * A,), device='cuda', generator=g).half()
            n = sum(h * w for h, w in shapes) * A
            sc = torch.full((B, n, 2), 9.0, device='cuda'); dl = torch.full((B, n, 4), 9.0, device='cuda')
            ops.rpn_head_fused(xs, w3, b3, w1, b1, A, sc, dl)
            ws, wd = [], []
            for x in xs:
                t = F.relu(F.conv2d(x.permute(0, 3, 1, 2).float(), w3.float(), b3.float(), 1, 1))
                o = F.conv2d(t, w1.float().reshape(6 * A, cout, 1, 1), b1.float()).permute(0, 2, 3, 1)
                ws.append(o[..., :2 * A].reshape(B, -1, 2)); wd.append(o[..., 2 * A:].reshape(B, -1, 4))
            ws, wd = torch.cat(ws, 1), torch.cat(wd, 1)
            assert float(ws.abs().max()) < 2048 and torch.equal(sc, ws) and torch.equal(dl, wd), ('rpn_head_fused', B, shapes, cin, cout, A)
        elif kind == 2:    # fused bottleneck tail
            cin, n3 = 64 * ri(1, 4), 64 * ri(1, 16)
            cm = (64, 128, 256)[ri(0, 2)]              # the 3x3 convolution's channels (conv2 / conv3 / conv4 of ResNet)
            x = sparse((B, H, W, cin), 4, 1, 1)
            w2 = sparse((cm, cin, 3, 3), 4, -2, 2).contiguous(memory_format=torch.channels_last)
            b2 = torch.randint(-3, 4, (cm,), device='cuda', generator=g).half()
            w3 = sparse((n3, cm), 10, -2, 2)
            b3 = torch.randint(-3, 4, (n3,), device='cuda', generator=g).half()
            r = torch.randint(-4, 5, (B, H, W, n3), device='cuda', generator=g).half() if case & 4 else None
            got = ops.conv3x3_conv1x1_f16(x, w2, b2, w3, b3, residual=r, relu=True)
            t = F.relu(F.conv2d(x.permute(0, 3, 1, 2).float(), w2.float(), b2.float(), 1, 1))
            o = F.conv2d(t, w3.float().reshape(n3, cm, 1, 1), b3.float()).permute(0, 2, 3, 1)
            want = F.relu(o + r.float() if r is not None else o)
            assert float(want.abs().max()) < 2048 and torch.equal(got.float(), want), ('block tail', B, H, W, cin, cm, n3)
        else:              # stem
            H2, W2 = ri(7, 140), ri(7, 180)
            img = torch.randint(-3, 4, (B, H2, W2, 3), device='cuda', generator=g).to(torch.float16 if case & 4 else torch.float32)
            w = torch.randint(-2, 3, (64, 3, 7, 7), device='cuda', generator=g).half()
            b = torch.randint(-3, 4, (64,), device='cuda', generator=g).half()
            got = ops.stem_conv7_pool3(img, ops.stem_pack_weights(w), b)
            y = F.relu(F.conv2d(F.pad(img.permute(0, 3, 1, 2).float(), (3, 3, 3, 3)), w.float(), b.float(), 2, 0))
            want = F.max_pool2d(F.pad(y, (1, 1, 1, 1)), 3, 2).permute(0, 2, 3, 1)
            assert torch.equal(got.float(), want), ('stem', B, H2, W2)
    torch.cuda.synchronize()
    print('all %d cases OK' % a.cases)


if __name__ == '__main__':
    main()

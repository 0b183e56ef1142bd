#!/usr/bin/env python3
"""1x1 stride-1 convolutions of the ResNet-101-FPN dense path (NHWC float16, batch 8 at 800x1333) two ways:
(a) library convolution without bias + the fused HIP epilogue (ops.bias_act_), what the detectors do;
(b) ONE GEMM with the bias + ReLU in its epilogue (torch._addmm_activation -> hipBLASLt) on the [pixels, C] view.

    python tools/exp/conv1x1_gemm.py"""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tf_eager_object_detection_amd import ops


def timeit(fn, warm=5, reps=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    torch.backends.cudnn.benchmark = True
    B = 8
    # (h, w, cin, cout): conv2..conv5 block convolutions c1 (4f -> f) and c3 (f -> 4f), laterals, the RoI head's fcs aside
    shapes = [(200, 334, 256, 64), (200, 334, 64, 256), (100, 167, 512, 128), (100, 167, 128, 512),
              (50, 84, 1024, 256), (50, 84, 256, 1024), (25, 42, 2048, 512), (25, 42, 512, 2048)]
    for h, w, cin, cout in shapes:
        x = torch.randn(B, cin, h, w, device='cuda', dtype=torch.float16).contiguous(memory_format=torch.channels_last)
        wt = (torch.randn(cout, cin, 1, 1, device='cuda', dtype=torch.float16) * 0.05).contiguous(memory_format=torch.channels_last)
        bias = torch.randn(cout, device='cuda', dtype=torch.float16)
        w2 = wt.reshape(cout, cin).t().contiguous()            # [cin, cout]
        w2t = wt.reshape(cout, cin)                            # [cout, cin] (F.linear's layout)

        def conv_epi():
            y = F.conv2d(x, wt, None)
            ops.bias_act_(y.permute(0, 2, 3, 1), bias, None, True)
            return y

        def gemm_epi():
            x2 = x.permute(0, 2, 3, 1).reshape(-1, cin)
            return torch._addmm_activation(bias, x2, w2, use_gelu=False)

        def linear_relu():
            x2 = x.permute(0, 2, 3, 1).reshape(-1, cin)
            return torch.relu_(F.linear(x2, w2t, bias))

        ya = conv_epi().permute(0, 2, 3, 1).reshape(-1, cout).float()
        yb = gemm_epi().float()
        err = float((ya - yb).abs().max() / ya.abs().max().clamp_min(1e-6))
        ta, tb, tc = timeit(conv_epi), timeit(gemm_epi), timeit(linear_relu)
        flop = 2.0 * B * h * w * cin * cout
        print('%4dx%-4d %4d -> %-4d  conv+epilogue %7.1f us (%5.0f TF/s)   addmm_activation %7.1f us (%5.0f TF/s)   '
              'linear+relu_ %7.1f us   max rel diff %.1e' % (h, w, cin, cout, ta, flop / ta / 1e6, tb, flop / tb / 1e6, tc, err))


if __name__ == '__main__':
    main()

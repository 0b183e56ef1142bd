#!/usr/bin/env python3
"""Round 4: what the second phase of the conv4 fused tail waits for: the product, without the shortcut (residual=None), and
with a diagnostic library whose output stores are dropped by the range check (ODET_LIB_PATH=tools/exp/libodet_tail_nostore.so).
    python tools/r04/tail_phase2_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tools._diag            # (ODET_LIB_PATH selects a diagnostic build: tools only, the product reads no environment)
import torch
from tf_eager_object_detection_amd import ops
B, H, W, cm, n3 = 30, 50, 84, 256, 1024
def mk():
    x = torch.randn(B, H, W, cm, device='cuda').half()
    r = torch.randn(B, H, W, n3, device='cuda').half()
    return x, r, torch.empty_like(r)
w2 = (torch.randn(cm, cm, 3, 3, device='cuda') * 0.05).half().contiguous(memory_format=torch.channels_last)
b2 = torch.randn(cm, device='cuda').half()
w3 = (torch.randn(n3, cm, device='cuda') * 0.05).half()
b3 = torch.randn(n3, device='cuda').half()
sets = [mk() for _ in range(2)]
def timed(fn, n=6):
    for s in sets: fn(s)
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n): fn(sets[i % 2])
    e.record(); e.synchronize()
    return a.elapsed_time(e) * 1e3 / n
print('lib', os.environ.get('ODET_LIB_PATH', 'product'))
print('  tail with shortcut    %.1f us' % timed(lambda s: ops.conv3x3_conv1x1_f16(s[0], w2, b2, w3, b3, residual=s[1], relu=True, out=s[2])))
print('  tail without shortcut %.1f us' % timed(lambda s: ops.conv3x3_conv1x1_f16(s[0], w2, b2, w3, b3, residual=None, relu=True, out=s[2])))
y2 = torch.empty(B, H, W, cm, device='cuda', dtype=torch.float16)
print('  3x3 alone (writes t)  %.1f us' % timed(lambda s: ops.conv3x3_f16(s[0], w2, b2, relu=True, out=y2)))

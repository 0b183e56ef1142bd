#!/usr/bin/env python3
"""Round 4: what would de-synchronising the two phases of the fused bottleneck tail be worth?  Two streams restricted to
complementary halves of the CUs (hipExtStreamCreateWithCUMask, alternating CUs), each running the conv4 tail on half of a
30-image batch: both at once (their matrix phases and their memory phases coincide, as inside one launch) against the
second stream delayed by about half a launch.    python tools/r04/tail_desync_probe.py"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tf_eager_object_detection_amd import ops

hip = C.CDLL('libamdhip64.so')
hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]


def masked_stream(pattern):
    h = C.c_void_p()
    mask = (C.c_uint32 * 8)(*([pattern] * 8))
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(h), 8, mask)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(h.value)


B, H, W, cm, n3 = 15, 50, 84, 256, 1024
def mk():
    x = torch.randn(B, H, W, cm, device='cuda').half()
    r = torch.randn(B, H, W, n3, device='cuda').half()
    return x, r, torch.empty_like(r)
w2 = (torch.randn(cm, cm, 3, 3, device='cuda') * 0.05).half().contiguous(memory_format=torch.channels_last)
b2 = torch.randn(cm, device='cuda').half()
w3 = (torch.randn(n3, cm, device='cuda') * 0.05).half()
b3 = torch.randn(n3, device='cuda').half()
sets = [mk() for _ in range(4)]
junk = torch.empty(1 << 26, device='cuda')


def tail(s):
    ops.conv3x3_conv1x1_f16(s[0], w2, b2, w3, b3, residual=s[1], relu=True, out=s[2])


def run(sa, sb, delay_elems, reps=6):
    """reps x (tail on stream sa, tail on stream sb after a filler of delay_elems elements) -> us per pair"""
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    cur = torch.cuda.current_stream()
    a.record(cur)
    sa.wait_stream(cur); sb.wait_stream(cur)
    with torch.cuda.stream(sb):
        if delay_elems:
            junk[:delay_elems].fill_(1.0)                     # (a memory-bound filler: shifts stream b's launches)
    for i in range(reps):
        with torch.cuda.stream(sa):
            tail(sets[(2 * i) % 4])
        with torch.cuda.stream(sb):
            tail(sets[(2 * i + 1) % 4])
    cur.wait_stream(sa); cur.wait_stream(sb)
    e.record(cur); e.synchronize()
    return a.elapsed_time(e) * 1e3 / reps


res = {}
even, odd = masked_stream(0x55555555), masked_stream(0xAAAAAAAA)
full_a, full_b = torch.cuda.Stream(), torch.cuda.Stream()
for name, sa, sb in (('two unmasked streams', full_a, full_b), ('complementary CU halves', even, odd)):
    for d in (0, 1 << 20, 1 << 22, 1 << 23, 1 << 24, 3 << 23, 1 << 25):
        run(sa, sb, d, 2)
        t = min(run(sa, sb, d) for _ in range(3))
        res['%s, filler %d MB' % (name, d * 4 >> 20)] = round(t, 1)
        print('%-28s filler %4d MB: %7.1f us per pair of 15-image tails' % (name, d * 4 >> 20, t), flush=True)
one = torch.cuda.Stream()
t = min(run(one, one, 0) for _ in range(3))
print('one stream, the two tails one after the other: %.1f us per pair' % t)
res['one stream, serial'] = round(t, 1)
os.makedirs('gpurun_out', exist_ok=True)
json.dump(res, open('gpurun_out/r04_tail_desync.json', 'w'), indent=1)

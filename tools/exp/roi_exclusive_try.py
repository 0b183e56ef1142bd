"""Experiment: the three stream groups of the bench fall into lockstep (all in their RoI kernels, then all in
their small kernels).  Here the RoI launches are serialised across the groups with HIP events (one RoI kernel
at a time at full speed, the other groups' small kernels beside it); driven from one Python thread through
odet_fpn_step_enqueue_batch with stage masks."""
import sys, time, ctypes as C, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tf_eager_object_detection_amd import _lib
from tf_eager_object_detection_amd.pipeline import FpnStreamPool, synthetic_fpn_inputs
S, B = int(sys.argv[1]) if len(sys.argv) > 1 else 3, 8
mode = sys.argv[2] if len(sys.argv) > 2 else 'exclusive'
shape = (800, 1333)
host, dev = synthetic_fpn_inputs(shape, 21, 1000, 256, seed=1234)
pool = FpnStreamPool(S, shape, 21, 1000, 256, batch=B)
gen = torch.Generator(device='cuda'); gen.manual_seed(1)
for k in range(pool.n):
    feats = dev['feats'] if k == 0 else [torch.randn(f.shape, device='cuda', generator=gen) for f in dev['feats']]
    pool.bind(k, dev['rpn_logits'], dev['rpn_deltas'], feats, dev['cls_scores'], dev['cls_deltas'])
lib = _lib.lib()
streams = pool._group_streams
def enqueue(g, stages):
    _lib.check(lib.odet_fpn_step_enqueue_batch(pool._groups[g], B, stages))
events = [torch.cuda.Event() for _ in range(S)]
def round_exclusive(first):
    for g in range(S):
        st = streams[g]
        enqueue(g, 1)
        prev = events[(g - 1) % S]
        if not (first and g == 0):
            st.wait_event(prev)
        enqueue(g, 2)
        events[g].record(st)
        enqueue(g, 4)
def round_plain(first):
    for g in range(S):
        enqueue(g, 7)
sec = [torch.cuda.Event() for _ in range(S)]
def round_mutex(first):
    """the small sections (detect of the previous batch + proposals of the next) of the groups never overlap"""
    for g in range(S):
        st = streams[g]
        if not (first and g == 0):
            st.wait_event(sec[(g - 1) % S])
        if not first:
            enqueue(g, 4)
        enqueue(g, 1)
        sec[g].record(st)
        enqueue(g, 2)
hi = [torch.cuda.Stream(priority=-1) for _ in range(S)]
lo = [torch.cuda.Stream(priority=0) for _ in range(S)]
e1 = [torch.cuda.Event() for _ in range(S)]
e2 = [torch.cuda.Event() for _ in range(S)]
def enqueue_on(g, stages, stream):
    for k in range(g * B, (g + 1) * B):
        pool.steps[k].stream = stream.cuda_stream
    enqueue(g, stages)
def round_prio(first):
    """small kernels (proposals, detect) on a high-priority stream per group, the RoI launch on a normal one"""
    for g in range(S):
        enqueue_on(g, 1, hi[g]); e1[g].record(hi[g])
        lo[g].wait_event(e1[g]); enqueue_on(g, 2, lo[g]); e2[g].record(lo[g])
        hi[g].wait_event(e2[g]); enqueue_on(g, 4, hi[g])
def round_plain2(first):
    for g in range(S):
        enqueue_on(g, 7, lo[g])
fn = {'exclusive': round_exclusive, 'plain': round_plain, 'mutex': round_mutex, 'prio': round_prio, 'plain2': round_plain2}[mode]
for i in range(10): fn(i == 0)
torch.cuda.synchronize()
t0 = time.perf_counter(); R = 80
for i in range(R): fn(False)
torch.cuda.synchronize()
el = time.perf_counter() - t0
print('%s S=%d: %.0f img/s (%.1f us/img)' % (mode, S, R * S * B / el, el / (R * S * B) * 1e6), 'nms_done', int(pool.slots[0].nms_done.item()))
pool.close()

import sys, time, torch
sys.path.insert(0, "/root/repo")
from tf_eager_object_detection_amd.pipeline import FpnHotPath, synthetic_fpn_inputs
host, dev = synthetic_fpn_inputs((800, 1333), 21, 1000, 256, seed=1234)
for S in (1, 2, 3, 4, 6, 8):
    hots = [FpnHotPath((800, 1333), 21, 1000, 256) for _ in range(S)]
    streams = [torch.cuda.Stream() for _ in range(S)]
    def step(i):
        k = i % S
        with torch.cuda.stream(streams[k]):
            h = hots[k]
            h.stage_proposals(dev['rpn_logits'], dev['rpn_deltas'])
            h.stage_roi(dev['feats'])
            h.stage_detect(dev['cls_scores'], dev['cls_deltas'])
    for i in range(4 * S): step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    N = 400
    for i in range(N): step(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ok = all(int(h.nms_done.item()) == 1 and int(h.det_count.item()) == 50 for h in hots)
    print('streams %d: %.1f us/image, %.0f img/s ok=%s' % (S, dt / N * 1e6, N / dt, ok))
# host-only cost of a step (no GPU wait): enqueue time
t0 = time.perf_counter()
for i in range(200): step(i)
t1 = time.perf_counter()
torch.cuda.synchronize()
print('host enqueue us/step', (t1 - t0) / 200 * 1e6)

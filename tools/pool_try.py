import sys, time, torch
sys.path.insert(0, '/root/repo')
from tf_eager_object_detection_amd.pipeline import FpnHotPath, FpnStreamPool, synthetic_fpn_inputs
host, dev = synthetic_fpn_inputs((800, 1333), 21, 1000, 256, seed=1234)
ref = FpnHotPath((800, 1333), 21, 1000, 256)
ref.step(dev['rpn_logits'], dev['rpn_deltas'], dev['feats'], dev['cls_scores'], dev['cls_deltas'])
torch.cuda.synchronize()
want = ref.record.clone()
wantf = ref.roi_features.clone()
for S in (2, 3, 4, 6, 8):
    pool = FpnStreamPool(S, (800, 1333), 21, 1000, 256)
    for k in range(S):
        pool.bind(k, dev['rpn_logits'], dev['rpn_deltas'], dev['feats'], dev['cls_scores'], dev['cls_deltas'])
    for i in range(4 * S): pool.submit()
    pool.wait(); torch.cuda.synchronize()
    N = 800
    t0 = time.perf_counter()
    for i in range(N): pool.submit()
    t1 = time.perf_counter()
    pool.wait()
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ok = all(torch.equal(h.record, want) and torch.equal(h.roi_features, wantf) and int(h.nms_done.item()) == 1 for h in pool.slots)
    print('pool streams %d: %.1f us/image, %.0f img/s (submit %.1f us, enqueued after %.1f us/img) identical=%s'
          % (S, dt / N * 1e6, N / dt, (t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6, ok))
    pool.close()

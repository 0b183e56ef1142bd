import sys, time, torch
sys.path.insert(0, "/root/repo")
from tf_eager_object_detection_amd.pipeline import FpnHotPath, FpnStreamPool, synthetic_fpn_inputs
host, dev = synthetic_fpn_inputs((800, 1333), 21, 1000, 256, seed=1234)
ref = FpnHotPath((800, 1333), 21, 1000, 256)
ref.step(dev['rpn_logits'], dev['rpn_deltas'], dev['feats'], dev['cls_scores'], dev['cls_deltas'])
torch.cuda.synchronize()
want = ref.record.clone()
wantf = ref.roi_features.clone()
for S, B in ((8, 1), (1, 8), (2, 8), (3, 8), (2, 4), (4, 4), (4, 2)):
    pool = FpnStreamPool(S, (800, 1333), 21, 1000, 256, batch=B)
    for k in range(pool.n):
        pool.bind(k, dev['rpn_logits'], dev['rpn_deltas'], dev['feats'], dev['cls_scores'], dev['cls_deltas'])
    for i in range(4 * S): pool.submit_group()
    pool.wait(); torch.cuda.synchronize()
    N = 200
    t0 = time.perf_counter()
    for i in range(N): pool.submit_group()
    pool.wait()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ok = all(torch.equal(h.record, want) and torch.equal(h.roi_features, wantf) and int(h.nms_done.item()) == 1 for h in pool.slots)
    print('streams %d x batch %d: %.1f us/image, %.0f img/s identical=%s' % (S, B, dt / (N * B) * 1e6, N * B / dt, ok))
    pool.close()

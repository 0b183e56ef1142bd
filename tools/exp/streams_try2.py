import sys, time, threading, torch
sys.path.insert(0, "/root/repo")
from tf_eager_object_detection_amd.pipeline import FpnHotPath, synthetic_fpn_inputs
host, dev = synthetic_fpn_inputs((800, 1333), 21, 1000, 256, seed=1234)
def run_step(h):
    h.stage_proposals(dev['rpn_logits'], dev['rpn_deltas'])
    h.stage_roi(dev['feats'])
    h.stage_detect(dev['cls_scores'], dev['cls_deltas'])
for S in (3, 4, 6):
    hots = [FpnHotPath((800, 1333), 21, 1000, 256) for _ in range(S)]
    streams = [torch.cuda.Stream() for _ in range(S)]
    graphs = []
    for k in range(S):
        with torch.cuda.stream(streams[k]):
            for _ in range(3): run_step(hots[k])
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=streams[k]):
            run_step(hots[k])
        graphs.append(g)
    torch.cuda.synchronize()
    for i in range(4 * S): graphs[i % S].replay()
    torch.cuda.synchronize()
    N = 400
    t0 = time.perf_counter()
    for i in range(N): graphs[i % S].replay()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ok = all(int(h.nms_done.item()) == 1 and int(h.det_count.item()) == 50 for h in hots)
    print('graphs, streams %d: %.1f us/image, %.0f img/s (host enqueue %.1f us) ok=%s' % (S, dt / N * 1e6, N / dt, (t1 - t0) / N * 1e6, ok))
# threads: one host thread per stream, eager launches
for S in (4,):
    hots = [FpnHotPath((800, 1333), 21, 1000, 256) for _ in range(S)]
    streams = [torch.cuda.Stream() for _ in range(S)]
    N = 400
    def worker(k, n):
        with torch.cuda.stream(streams[k]):
            for _ in range(n): run_step(hots[k])
    ths = [threading.Thread(target=worker, args=(k, 8)) for k in range(S)]
    [t.start() for t in ths]; [t.join() for t in ths]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ths = [threading.Thread(target=worker, args=(k, N // S)) for k in range(S)]
    [t.start() for t in ths]; [t.join() for t in ths]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print('threads, streams %d: %.1f us/image, %.0f img/s' % (S, dt / N * 1e6, N / dt))

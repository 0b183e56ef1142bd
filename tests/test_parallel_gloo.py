"""World-size-2 test of the image-parallel harness on CPU (gloo): sharding, record packing, the ONE
all-gather per batch, unpacking.  The GPU path differs only in the backend ("nccl" = RCCL) and in
odet_pack_detections producing the record on the device."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tf_eager_object_detection_amd import parallel

MAX_DET = 50


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_detections(image_id):
    """Deterministic padded post-ops outputs of one 'image' (count depends on the id, 0 allowed)."""
    rng = np.random.default_rng(100 + image_id)
    m = [0, 7, 50, 23][image_id % 4]
    boxes = np.zeros((MAX_DET, 4), np.float32)
    labels = np.zeros(MAX_DET, np.int32)
    scores = np.zeros(MAX_DET, np.float32)
    boxes[:m] = rng.uniform(0, 800, (m, 4)).astype(np.float32)
    labels[:m] = rng.integers(1, 21, m).astype(np.int32)
    scores[:m] = np.sort(rng.uniform(0, 1, m).astype(np.float32))[::-1]
    return boxes, labels, scores, m


def _worker(rank, world, port, num_images, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        mine = parallel.shard_images(num_images, rank, world)
        got = {}
        steps = (num_images + world - 1) // world
        for s in range(steps):
            if s < len(mine):
                b, l, sc, m = _fake_detections(mine[s])
                rec = parallel.pack_detections(torch.from_numpy(b), torch.from_numpy(l), torch.from_numpy(sc),
                                               torch.tensor([m], dtype=torch.int32), MAX_DET)
            else:      # ragged tail: this rank has no image in the last step, it contributes an empty record
                z = torch.zeros((MAX_DET, 4))
                rec = parallel.pack_detections(z, torch.zeros(MAX_DET, dtype=torch.int32), torch.zeros(MAX_DET),
                                               torch.tensor([0], dtype=torch.int32), MAX_DET)
            allrec = parallel.all_gather_detections(rec)          # ONE collective per step
            assert allrec.shape == (world, MAX_DET * 6 + 1)
            for r in range(world):
                img = r + s * world
                if img < num_images:
                    got[img] = [t.numpy().copy() for t in parallel.unpack_detections(allrec[r], MAX_DET)]
        q.put((rank, mine, got))
    finally:
        dist.destroy_process_group()


def _run(world, num_images):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, num_images, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return results


def test_shard_images_round_robin():
    assert parallel.shard_images(8, 0, 2) == [0, 2, 4, 6]
    assert parallel.shard_images(8, 1, 2) == [1, 3, 5, 7]
    assert parallel.shard_images(5, 1, 2) == [1, 3]
    assert parallel.shard_images(1, 3, 8) == []
    cover = sorted(sum((parallel.shard_images(13, r, 4) for r in range(4)), []))
    assert cover == list(range(13))


def test_pack_unpack_roundtrip_cpu():
    b, l, s, m = _fake_detections(1)
    rec = parallel.pack_detections(torch.from_numpy(b), torch.from_numpy(l), torch.from_numpy(s),
                                   torch.tensor([m], dtype=torch.int32), MAX_DET)
    assert rec.shape == (MAX_DET * 6 + 1,)
    body = rec[:MAX_DET * 6].view(MAX_DET, 6)
    assert float(rec[-1]) == m
    assert torch.all(body[m:, 4] == -1.0) and torch.all(body[m:, :4] == 0)
    ub, ul, us = parallel.unpack_detections(rec, MAX_DET)
    np.testing.assert_array_equal(ub.numpy(), b[:m])
    np.testing.assert_array_equal(ul.numpy(), l[:m])
    np.testing.assert_array_equal(us.numpy(), s[:m])


def test_all_gather_detections_world2_gloo():
    num_images = 5                       # ragged: rank 1 idles in the last step
    results = _run(2, num_images)
    assert sorted(r[0] for r in results) == [0, 1]
    for rank, mine, got in results:
        assert mine == list(range(rank, num_images, 2))
        assert sorted(got) == list(range(num_images))           # every rank sees every image
        for img in range(num_images):
            b, l, s, m = _fake_detections(img)
            gb, gl, gs = got[img]
            np.testing.assert_array_equal(gb, b[:m])
            np.testing.assert_array_equal(gl, l[:m])
            np.testing.assert_array_equal(gs, s[:m])


# ---- the exchange bench.py times: stream groups of B images, [S, world, B, rec_len], ragged tail ----------------
def _group_worker(rank, world, port, S, B, rounds, num_images, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        rec_len = MAX_DET * 6 + 1
        ex = parallel.GroupExchange(S, B, rec_len, 'cpu')
        mine = parallel.shard_images(num_images, rank, world)
        got = {}
        k = 0                                     # next local image
        for rd in range(rounds):
            for g in range(S):
                block = torch.zeros((B, rec_len))
                valid = 0
                for j in range(B):
                    if k + j < len(mine):
                        b, l, sc, m = _fake_detections(mine[k + j])
                        block[j] = parallel.pack_detections(torch.from_numpy(b), torch.from_numpy(l),
                                                            torch.from_numpy(sc), torch.tensor([m], dtype=torch.int32),
                                                            MAX_DET)
                        valid += 1
                    else:
                        block[j] = 123.0          # stale slot contents: must never reach another rank
                out = ex.gather(g, block, valid=valid)
                assert out.shape == (world, B, rec_len)
                for r in range(world):
                    for j in range(B):
                        local = k + j             # local index on rank r
                        img = r + local * world
                        cnt = int(out[r, j, -1].item())
                        if img < num_images:
                            got[img] = [t.numpy().copy() for t in parallel.unpack_detections(out[r, j], MAX_DET)]
                        else:
                            assert cnt == 0 and torch.all(out[r, j, :-1].view(-1, 6)[:, 4] == -1.0)
                k += B
        q.put((rank, got))
    finally:
        dist.destroy_process_group()


def test_group_exchange_world2_gloo_ragged():
    world, S, B, rounds = 2, 3, 4, 2
    num_images = 41                   # 2 ranks x 24 slots; rank 1's last group is ragged, the one after empty
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_group_worker, args=(r, world, port, S, B, rounds, num_images, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, got in results:
        assert sorted(got) == list(range(num_images))           # every rank ends with every image's record
        for img in range(num_images):
            b, l, s, m = _fake_detections(img)
            gb, gl, gs = got[img]
            np.testing.assert_array_equal(gb, b[:m])
            np.testing.assert_array_equal(gl, l[:m])
            np.testing.assert_array_equal(gs, s[:m])


def test_group_exchange_single_rank_cpu():
    ex = parallel.GroupExchange(2, 3, MAX_DET * 6 + 1, 'cpu')
    block = torch.arange(3 * (MAX_DET * 6 + 1), dtype=torch.float32).view(3, -1)
    out = ex.gather(1, block, valid=2)
    assert out.shape == (1, 3, MAX_DET * 6 + 1)
    assert torch.equal(out[0, :2], block[:2])
    assert float(out[0, 2, -1]) == 0.0 and torch.all(out[0, 2, :-1].view(-1, 6)[:, 4] == -1.0)

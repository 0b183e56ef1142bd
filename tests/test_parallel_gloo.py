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

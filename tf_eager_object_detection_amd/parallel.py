"""Image-parallel multi-GPU harness: one process per GPU (torch.distributed, backend "nccl" =
RCCL over xGMI on ROCm; "gloo" for the CPU tests), images sharded round-robin over ranks,
ONE all-gather of fixed-size detection records per batch.

The reference has no multi-GPU code at all (README "multi gpu support" is an unchecked TODO;
batch is hard-wired to 1: model/roi_pooling.py:28,66,152).  The path shards naturally by image
-- no stage has cross-image state -- so there is no data-path collective inside an image; the
only exchange is the final record all-gather, which is latency-bound (<= 7.2 KB per image).

Record layout per image: float32 [max_det, 6] = (x1, y1, x2, y2, score, label), padded rows have
score = -1; plus the valid count.  Packed as one float32 [max_det*6 + 1] vector so a batch costs
exactly one collective.
"""
import torch
import torch.distributed as dist


def shard_images(num_images, rank, world_size):
    """Image indices owned by `rank`: r, r+G, r+2G, ... (weak scaling: per-GPU work is fixed when the
    global batch grows with the number of GPUs)."""
    return list(range(rank, num_images, world_size))


def pack_detections(boxes, labels, scores, count, max_det, out=None):
    """Padded post-ops outputs (+ device count) -> float32 [max_det*6+1] record vector.  GPU tensors:
    one HIP launch (odet_pack_detections), no host sync.  CPU tensors (gloo tests): torch ops."""
    if boxes.is_cuda:
        from . import _lib as L
        rec = out if out is not None else torch.empty(max_det * 6 + 1, dtype=torch.float32, device=boxes.device)
        L.check(L.lib().odet_pack_detections(L.dptr(boxes), L.dptr(labels), L.dptr(scores), L.dptr(count),
                                             boxes.shape[0], int(max_det), L.dptr(rec), L.stream()))
        return rec
    m = min(int(count.reshape(-1)[0]), boxes.shape[0], max_det)
    rec = torch.zeros(max_det * 6 + 1, dtype=torch.float32)
    body = rec[:max_det * 6].view(max_det, 6)
    body[:, 4] = -1.0
    body[:m, 0:4] = boxes[:m]
    body[:m, 4] = scores[:m]
    body[:m, 5] = labels[:m].to(torch.float32)
    rec[max_det * 6] = float(m)
    return rec


def unpack_detections(rec, max_det):
    """record vector -> (boxes [M,4], labels int32 [M], scores [M]) with M = stored count (host sync)."""
    m = int(rec[max_det * 6].item())
    body = rec[:max_det * 6].view(max_det, 6)[:m]
    return body[:, 0:4], body[:, 5].to(torch.int32), body[:, 4]


def all_gather_detections(rec, group=None):
    """One collective per batch: every rank receives every rank's record vector -> [world, len]."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return rec[None]
    out = torch.empty((world, rec.numel()), dtype=rec.dtype, device=rec.device)
    if rec.is_cuda:
        dist.all_gather_into_tensor(out, rec.contiguous(), group=group)
    else:
        parts = [torch.empty_like(rec) for _ in range(world)]     # gloo: list form
        dist.all_gather(parts, rec.contiguous(), group=group)
        out = torch.stack(parts, 0)
    return out


class GroupExchange:
    """The exchange of the throughput arrangement (bench.py, FpnStreamPool): every rank keeps S stream groups of B
    images in flight; when a group's launches are enqueued, the B fixed-size records of that group of EVERY rank go
    through ONE all-gather (concatenation form) into ``gathered[g]`` = [world, B, rec_len].

    GPU tensors: the collective runs on a communication stream that waits (on the device) for the group's stream;
    the group's stream only waits for the small staging copy, so its next images may overwrite the records while
    the collective is still in flight.  CPU tensors (gloo tests): the same calls without streams.

    A group with fewer than B valid images (ragged tail of a finite image list) passes ``valid``: the records of the
    missing images are sent as empty records (count 0, scores -1), so every rank still contributes B records."""

    def __init__(self, n_groups, batch, rec_len, device, group=None, force_collective=False):
        self.S, self.B, self.rec_len = int(n_groups), int(batch), int(rec_len)
        self.group = group
        # (a one-rank process group still issues the collective: rehearsal of the RCCL call path on a 1-GPU box)
        self.force_collective = bool(force_collective) and dist.is_initialized()
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        dev = torch.device(device)
        self.cuda = dev.type == 'cuda'
        self.gathered = torch.zeros((self.S, self.world, self.B, self.rec_len), dtype=torch.float32, device=dev)
        self.staging = torch.zeros((self.S, self.B, self.rec_len), dtype=torch.float32, device=dev)
        self.comm = torch.cuda.Stream(device=dev) if self.cuda else None
        empty = torch.zeros(self.rec_len, dtype=torch.float32)
        empty[:self.rec_len - 1].view(-1, 6)[:, 4] = -1.0
        self.empty = empty.to(dev)

    def _stage(self, g, records, valid):
        self.staging[g].copy_(records)
        if valid is not None and valid < self.B:
            self.staging[g, valid:] = self.empty

    def gather(self, g, records, producer_stream=None, valid=None):
        """records: [B, rec_len] block of group g (written on ``producer_stream``).  Returns gathered[g]
        ([world, B, rec_len]; on the GPU it is complete once the communication stream has run)."""
        if not self.cuda:
            self._stage(g, records, valid)
            if self.world == 1:
                self.gathered[g, 0].copy_(self.staging[g])
            else:
                parts = [torch.empty_like(self.staging[g]) for _ in range(self.world)]
                dist.all_gather(parts, self.staging[g].contiguous(), group=self.group)
                self.gathered[g].copy_(torch.stack(parts, 0))
            return self.gathered[g]
        st = producer_stream if producer_stream is not None else torch.cuda.current_stream()
        self.comm.wait_stream(st)
        with torch.cuda.stream(self.comm):
            self._stage(g, records, valid)
            copied = torch.cuda.Event()
            copied.record(self.comm)
            if self.world == 1 and not self.force_collective:
                self.gathered[g, 0].copy_(self.staging[g])
            elif dist.get_backend(self.group) == 'gloo':
                # rehearsal of the multi-rank path on a box with fewer GPUs than ranks: gloo moves host memory
                host = self.staging[g].cpu()                      # (synchronises the communication stream)
                parts = [torch.empty_like(host) for _ in range(self.world)]
                dist.all_gather(parts, host, group=self.group)
                self.gathered[g].copy_(torch.stack(parts, 0))
            else:
                dist.all_gather_into_tensor(self.gathered[g].view(self.world * self.B, self.rec_len), self.staging[g],
                                            group=self.group)
        st.wait_event(copied)            # the group's next images may overwrite its records from here on
        return self.gathered[g]

    def synchronize(self):
        if self.cuda:
            self.comm.synchronize()

"""Batched-frames mode across the GPUs of one node.

Frame pairs are independent, so the batch is cut into contiguous shards, one per
rank (one process per GPU); there is no exchange step in the algorithm and no
data-path collective. The only communication is the gather of the per-pair flow
vectors (a few KB per pair) with one ``all_gather`` -- RCCL over xGMI when the
process group is ``nccl``, gloo in the CPU tests. The reference itself is single
process / single device ("Parallelism strategies: none").
"""
from __future__ import annotations

from typing import Callable


def shard_bounds(n_pairs: int, rank: int, world: int) -> tuple[int, int]:
    """Pairs [lo, hi) owned by ``rank``: contiguous shards of ceil(n/world), last ones may be short/empty."""
    if world < 1 or not (0 <= rank < world) or n_pairs < 0:
        raise ValueError("bad shard arguments")
    per = -(-n_pairs // world)
    lo = min(n_pairs, rank * per)
    return lo, min(n_pairs, lo + per)


def gather_results(local, n_pairs: int, group=None):
    """All-gather per-rank result slabs ``[shard, ...]`` into ``[n_pairs, ...]`` on every rank.

    Shards are padded to ceil(n/world) rows so a single fixed-size ``all_gather_into_tensor``
    (one RCCL call) carries everything; padding rows are dropped afterwards.
    """
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        assert local.shape[0] == n_pairs
        return local
    world = dist.get_world_size(group)
    per = -(-n_pairs // world)
    tail = tuple(local.shape[1:])
    slab = local
    if local.shape[0] != per:
        slab = torch.zeros((per,) + tail, dtype=local.dtype, device=local.device)
        slab[: local.shape[0]] = local
    slab = slab.contiguous()
    out = torch.empty((world * per,) + tail, dtype=local.dtype, device=local.device)
    if dist.get_backend(group) == "gloo":
        parts = [torch.empty_like(slab) for _ in range(world)]
        dist.all_gather(parts, slab, group=group)
        out = torch.cat(parts, dim=0)
    else:
        dist.all_gather_into_tensor(out, slab, group=group)
    return out[:n_pairs]


class GatherBufferInUse(RuntimeError):
    """A gathered buffer is about to be overwritten while the consumer it was handed to has not released it."""


class Gathered:
    """What AsyncGather.submit() hands out: the gathered tensor of one batch plus its hand-over protocol.
    ``wait()`` -- the gather has landed (torch's current stream is ordered behind it); ``tensor`` -- [n_pairs, ...];
    ``done()`` -- the consumer is finished: work it queued on the current stream so far is what the buffer's next
    gather will wait for. A buffer whose consumer never called done() is not re-used (GatherBufferInUse)."""

    def __init__(self, tensor, work):
        self.tensor, self._work, self._event, self._done = tensor, work, None, False

    def wait(self):
        if self._work is not None:
            self._work.wait()
        return self.tensor

    def done(self):
        import torch

        if not self._done:
            if self.tensor.is_cuda:  # (CPU tensors -- the gloo tests -- are consumed synchronously: nothing to wait for)
                self._event = torch.cuda.Event()
                self._event.record()
            self._done = True


class AsyncGather:
    """Double-buffered, non-blocking gather for back-to-back batches: the all-gather of batch i (RCCL, its own
    stream) overlaps the kernels of batch i+1; a local buffer is waited for only right before it is reused, and a
    gathered buffer is overwritten only after the consumer it was handed to has released it (Gathered.done())."""

    def __init__(self, shard_shape, dtype, device, n_pairs: int, group=None, depth: int = 2):
        import torch
        import torch.distributed as dist

        self.group, self.n_pairs, self.depth = group, n_pairs, depth
        self.world = dist.get_world_size(group)
        self.per = -(-n_pairs // self.world)
        assert shard_shape[0] == self.per, "AsyncGather needs equal shards (pad the last one)"
        self.local = [torch.empty(shard_shape, dtype=dtype, device=device) for _ in range(depth)]
        self.full = [torch.empty((self.world * self.per,) + tuple(shard_shape[1:]), dtype=dtype, device=device)
                     for _ in range(depth)]
        self.work = [None] * depth
        self.handed = [None] * depth
        self.i = 0

    def slot(self):
        """Local result buffer for the next batch (waits for the gather that last read it)."""
        k = self.i % self.depth
        if self.work[k] is not None:
            self.work[k].wait()
            self.work[k] = None
        return self.local[k]

    def submit(self) -> Gathered:
        import torch
        import torch.distributed as dist

        k = self.i % self.depth
        prev = self.handed[k]
        if prev is not None:
            if not prev._done:
                raise GatherBufferInUse(f"gathered buffer {k} (batch {self.i - self.depth}) is still held by its consumer: "
                                        "call done() on the handle submit() returned before this buffer comes round again")
            # the collective is ordered behind the current stream: make that stream wait for the consumer's last read
            if prev._event is not None:
                torch.cuda.current_stream().wait_event(prev._event)
        if self.full[k].is_cuda:
            self.work[k] = dist.all_gather_into_tensor(self.full[k], self.local[k], group=self.group, async_op=True)
        else:  # gloo on CPU tensors has no all_gather_into_tensor: gather into row views of the same buffer
            parts = list(self.full[k].split(self.per, dim=0))
            self.work[k] = dist.all_gather(parts, self.local[k], group=self.group, async_op=True)
        self.handed[k] = Gathered(self.full[k][: self.n_pairs], self.work[k])
        self.i += 1
        return self.handed[k]

    def drain(self):
        for k in range(self.depth):
            if self.work[k] is not None:
                self.work[k].wait()
                self.work[k] = None


def run_sharded(process_shard: Callable, n_pairs: int, rank: int, world: int, group=None):
    """``process_shard(lo, hi) -> tensor [hi-lo, ...]`` on this rank's shard, then gather."""
    lo, hi = shard_bounds(n_pairs, rank, world)
    local = process_shard(lo, hi)
    assert local.shape[0] == hi - lo
    return gather_results(local, n_pairs, group)

"""Synthetic generator (numpy == torch bytes), shard arithmetic, and the world_size-2 gather over gloo."""
import os
import socket

import numpy as np
import pytest
import torch

from mrs_optic_flow_amd import sharding, synth


def test_numpy_and_torch_generators_agree_bytewise():
    for blur in (False, True):
        cn, pn, sn, kn = synth.batch_np(24, 40, 56, 5, blur=blur, k0=0)
        ct, pt, st_, kt = synth.batch_torch(24, 40, 56, 5, "cpu", blur=blur, k0=0, chunk=7)
        assert kn == kt and set(kn) == {"shift", "identical", "constant", "noisy"}
        assert (ct.numpy() == cn).all() and (pt.numpy() == pn).all() and (st_.numpy() == sn).all()


def test_planted_translation_moves_content():
    cur, prev = synth.pair_np(5, 48, 64, 3, -2)
    assert (cur[10:30, 10:30] == prev[12:32, 7:27]).all()  # cur(y,x) = prev(y-dy, x-dx)


@pytest.mark.parametrize("n,world", [(1024, 8), (10, 4), (3, 8), (0, 2), (7, 1)])
def test_shard_bounds_partition_the_batch(n, world):
    spans = [sharding.shard_bounds(n, r, world) for r in range(world)]
    covered = [k for lo, hi in spans for k in range(lo, hi)]
    assert covered == list(range(n))
    assert max(hi - lo for lo, hi in spans) == -(-n // world) if n else True


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_pairs, q):
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [os.path.dirname(here), here]
    import torch.distributed as dist

    import oracle_lib as O
    from mrs_optic_flow_amd import sharding, synth

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    h, w = 64, 96
    lay = O.fft_layout(w, h, 32, 2, 1, (3, 7), (40, 1))

    def process_shard(lo, hi):  # the oracle stands in for the HIP engine on CPU ranks
        cur, prev, _, _ = synth.batch_np(hi - lo, h, w, 4, k0=lo)
        out = np.zeros((hi - lo, 2, 2))
        for k in range(hi - lo):
            out[k], _ = O.fft_process(cur[k], prev[k], lay, 64)
        return torch.from_numpy(out)

    full = sharding.run_sharded(process_shard, n_pairs, rank, world)
    q.put((rank, full.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_pairs", [5, 8])
def test_two_rank_gather_equals_single_rank(n_pairs):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_pairs, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    import oracle_lib as O

    lay = O.fft_layout(96, 64, 32, 2, 1, (3, 7), (40, 1))
    cur, prev, _, _ = synth.batch_np(n_pairs, 64, 96, 4, k0=0)
    want = np.stack([O.fft_process(cur[k], prev[k], lay, 64)[0] for k in range(n_pairs)])
    for r in range(2):
        assert got[r].shape == want.shape
        assert np.array_equal(got[r], want, equal_nan=True)  # bit-identical to the 1-rank result


def _async_worker(rank, world, port, q):
    import os
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [os.path.dirname(here), here]
    import torch.distributed as dist

    from mrs_optic_flow_amd import sharding

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    per, steps = 3, 7
    ag = sharding.AsyncGather((per, 2, 2), torch.float64, torch.device("cpu"), per * world)
    seen = []
    for i in range(steps):
        buf = ag.slot()
        buf.fill_(100.0 * i + rank)  # this rank's "flow vectors" of batch i
        h = ag.submit()
        if i >= 1:  # consume batch i - 1 one step late, as an overlapped pipeline does
            prev_h.wait()
            seen.append(prev_h.tensor[:, 0, 0].clone())
            prev_h.done()
        prev_h = h
    prev_h.wait()
    seen.append(prev_h.tensor[:, 0, 0].clone())
    prev_h.done()
    ag.drain()
    # a consumer that never releases its buffer stops the pipeline loudly instead of being overwritten
    blocked = False
    try:
        for i in range(3):
            ag.slot().fill_(-1.0)
            ag.submit()
    except sharding.GatherBufferInUse:
        blocked = True
    q.put((rank, torch.stack(seen).numpy(), blocked))
    dist.barrier()
    dist.destroy_process_group()


def test_async_gather_two_ranks_gloo():
    """sharding.AsyncGather with world_size 2 (gloo, CPU tensors): every rank sees every rank's shard of every batch, in
    order, while consuming one step late (double buffering); a buffer still held by its consumer is not re-used."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_async_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, seen, blocked in got:
        assert blocked
        assert seen.shape == (7, 6)
        for i in range(7):
            assert list(seen[i]) == [100.0 * i + 0] * 3 + [100.0 * i + 1] * 3, (rank, i, seen[i])

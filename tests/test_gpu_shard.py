"""GPU tests of the batched-frames mode across devices: the torch.distributed path (two HIP ranks on one GPU, the
non-blocking RCCL gather in a 1-rank world) and the native shard group behind the C ABI (mof_shard_fft_* / mof_shard_bm_*,
csrc/mof_shard.hip) from a C++ host and through ctypes, G > 1 rehearsed on one device under MOF_SHARD_SHARE_DEVICE=1."""
import ctypes as C
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from mrs_optic_flow_amd import FastSpacedBMMethod, FftMethod, synth

pytestmark = pytest.mark.gpu
TOL = 1e-4
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_RANK_SCRIPT = r"""
import os, sys
sys.path[:0] = [{root!r}, os.path.join({root!r}, "tests")]
import numpy as np, torch, torch.distributed as dist
from mrs_optic_flow_amd import FftMethod, FastSpacedBMMethod, sharding, synth
rank, world, port, n_pairs, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{{port}}", rank=rank, world_size=world)
dev = torch.device("cuda", 0)   # both ranks share the one GPU of the box
h, w, n = 480, 752, 64
fm = FftMethod(sample_point_size=n, frame_shape=(h, w), grid=(8, 8), origin=(1, 1), stride=(98, 59))
bm = FastSpacedBMMethod(16, 16, 8, (h, w))
def fft_shard(lo, hi):
    cur, prev, _, _ = synth.batch_torch(hi - lo, h, w, n // 8, dev, k0=lo)
    r = fm.process_batch_device(cur, prev); torch.cuda.synchronize(); return r.cpu()
def bm_shard(lo, hi):
    cur, prev, _, _ = synth.batch_torch(hi - lo, h, w, 12, dev, k0=lo)
    dx, dy, mode = bm.process_batch_device(cur, prev); torch.cuda.synchronize()
    return torch.cat([dx.reshape(hi - lo, -1), dy.reshape(hi - lo, -1), mode], dim=1).cpu()
full = sharding.run_sharded(fft_shard, n_pairs, rank, world)
full_bm = sharding.run_sharded(bm_shard, n_pairs, rank, world)
np.savez(out, fft=full.numpy(), bm=full_bm.numpy())
dist.barrier(); dist.destroy_process_group()
"""
_GATHER_SCRIPT = r"""
import os, sys
sys.path.insert(0, {root!r})
import torch, torch.distributed as dist
from mrs_optic_flow_amd import sharding
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
ag = sharding.AsyncGather((8, 4, 2), torch.float64, dev, 8)
seen = []
for i in range(6):
    buf = ag.slot(); buf.fill_(float(i)); full = ag.submit()
    full.wait()                                   # consumer side: the gather of batch i has landed
    seen.append(float(full.tensor[3, 1, 0]))
    full.done()                                   # ... and the consumer is finished with it
ag.drain(); torch.cuda.synchronize()
assert seen == [0.0, 1.0, 2.0, 3.0, 4.0, 5.0], seen
# a consumer that never says done() blocks the re-use of its buffer loudly instead of being overwritten
h0 = None
try:
    for i in range(3):
        ag.slot().fill_(9.0); h = ag.submit(); h0 = h0 or h
    raise SystemExit("buffer re-used under a pending consumer")
except sharding.GatherBufferInUse:
    pass
h0.done()
x = torch.arange(6, dtype=torch.float64, device=dev).reshape(3, 2)
assert sharding.gather_results(x, 3).tolist() == x.tolist()
dist.barrier(); dist.destroy_process_group()
print("async gather ok")
"""
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("n_pairs", [37])
def test_two_rank_hip_engine_equals_single_rank(gpu, tmp_path, n_pairs):
    """the batch sharded over two ranks (fresh processes, gloo rendezvous on 127.0.0.1, both on the one
    GPU) and gathered with sharding.run_sharded gives the 1-rank result bit for bit -- c2 geometry for the FFT path,
    c3 geometry for the block scan; 37 pairs -> shards of 19 and 18 (ragged)."""
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT.format(root=ROOT))
    port = _free_port()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "2", str(port), str(n_pairs),
                               str(tmp_path / f"r{r}.npz")], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                              text=True) for r in range(2)]
    for p in procs:
        out, _ = p.communicate(timeout=600)
        assert p.returncode == 0, out[-3000:]
    h, w, n = 480, 752, 64
    fm = FftMethod(sample_point_size=n, frame_shape=(h, w), grid=(8, 8), origin=(1, 1), stride=(98, 59))
    cur, prev, _, _ = synth.batch_torch(n_pairs, h, w, n // 8, gpu, k0=0)
    want = fm.process_batch_device(cur, prev).cpu().numpy()
    bm = FastSpacedBMMethod(16, 16, 8, (h, w))
    cur, prev, _, _ = synth.batch_torch(n_pairs, h, w, 12, gpu, k0=0)
    dx, dy, mode = bm.process_batch_device(cur, prev)
    want_bm = torch.cat([dx.reshape(n_pairs, -1), dy.reshape(n_pairs, -1), mode], dim=1).cpu().numpy()
    for r in range(2):
        got = np.load(tmp_path / f"r{r}.npz")
        assert got["fft"].shape == want.shape and np.array_equal(got["fft"], want, equal_nan=True)
        assert np.array_equal(got["bm"], want_bm)


def test_async_gather_on_rccl_in_a_fresh_process(gpu):
    """sharding.AsyncGather over a 1-rank RCCL group (the only NCCL world a 1-GPU box allows), in a child process so
    that the test runner itself never initialises a process group."""
    script = _GATHER_SCRIPT.format(root=ROOT, port=_free_port())
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "async gather ok" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


def test_native_sharded_entry_from_a_cpp_host(gpu):
    """tests/cpp/test_shard.cpp: mof_shard_fft_* and (r05) mof_shard_bm_* -- one process, one engine and stream per device,
    ceil(B / G) contiguous shards, ONE in-place RCCL all-gather (explicit mof_shard_*_init_gather = ncclCommInitAll, then
    ncclAllGather through the run-time-bound librccl) -- with the devices this box has; every device's gathered result equals the
    single-engine result on the whole batch bit for bit (FFT vectors; block shifts and modes in one slab)."""
    binp = os.path.join(ROOT, "tests", "cpp", "test_shard")
    assert os.path.exists(binp), "tests/cpp/test_shard missing: run __graft_entry__.build()"
    for pairs in (37, 8):
        r = subprocess.run([binp, str(pairs)], capture_output=True, text=True, timeout=300,
                           env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
        assert r.returncode == 0 and f"shard ok {torch.cuda.device_count()} {pairs}" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


def test_native_sharded_entry_through_ctypes(gpu):
    """The same entry from Python's ctypes binding, without the gather (gather = 0 needs no RCCL) and with it."""
    import ctypes as C
    from mrs_optic_flow_amd import _capi

    lib = _capi.load()
    B, h, w = 11, 136, 200
    cur, prev, _, _ = synth.batch_np(B, h, w, 5, k0=9)
    fm = FftMethod(sample_point_size=64, frame_shape=(h, w), grid=(2, 2), origin=(3, 1), stride=(97, 59))
    tc, tp = torch.from_numpy(cur).to(gpu), torch.from_numpy(prev).to(gpu)
    want = fm.process_batch_device(tc, tp)
    grp = C.c_void_p()
    _capi.check(lib.mof_shard_fft_create(C.byref(fm.cfg), None, 1, C.byref(grp)))
    try:
        assert lib.mof_shard_fft_devices(grp) == 1
        for gather in (0, 1):
            out = torch.full((B, 4, 2), float("nan"), dtype=torch.float64, device=gpu)
            pc, pp, po = (C.c_void_p * 1)(tc.data_ptr()), (C.c_void_p * 1)(tp.data_ptr()), (C.c_void_p * 1)(out.data_ptr())
            torch.cuda.synchronize()
            if gather:  # r05: the gather's set-up is explicit -- the asynchronous call refuses to build communicators itself
                assert lib.mof_shard_fft_process_batch_device(grp, pc, tc.stride(0), pp, tp.stride(0), tc.stride(1), B, po, 1) == _capi.MOF_ERR_NOT_INIT
                _capi.check(lib.mof_shard_fft_init_gather(grp))
                assert lib.mof_shard_fft_gather_ready(grp) == 1
            _capi.check(lib.mof_shard_fft_process_batch_device(grp, pc, tc.stride(0), pp, tp.stride(0), tc.stride(1), B, po, gather))
            _capi.check(lib.mof_shard_fft_sync(grp))
            assert torch.equal(out, want), gather
    finally:
        lib.mof_shard_fft_destroy(grp)


@pytest.mark.parametrize("pairs,G", [(37, 2), (37, 4), (9, 4), (3, 4), (1, 2), (8, 2)])
def test_shard_group_with_several_shards_on_one_device(gpu, pairs, G):
    """`mof_shard_*_process_batch_device`'s G > 1 arithmetic -- slab i at i * slab, ragged last shards
    (37 over 4 -> 10 10 10 7), empty ones (9 over 4 -> 3 3 3 0; 3 over 4 -> 1 1 1 0; 1 over 2) -- executed for real: the rehearsal knob
    MOF_SHARD_SHARE_DEVICE=1 admits G shards on the one device with gather = 0, and tests/cpp/test_shard.cpp checks that every
    slab lands at its place bit-equal to the single-engine call and that nothing else of the buffer is written -- FftMethod vectors,
    and FastSpacedBMMethod's dx | dy | mode planes. The all-gather itself stays a 1-rank run (RCCL: one rank per device) until a
    multi-GPU node exists; a shared-device group refuses it (checked inside)."""
    binp = os.path.join(ROOT, "tests", "cpp", "test_shard")
    assert os.path.exists(binp), "tests/cpp/test_shard missing: run __graft_entry__.build()"
    r = subprocess.run([binp, "rehearse", str(pairs), str(G)], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, MOF_SHARD_SHARE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0 and f"rehearse ok {G} {pairs}" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


def test_block_matching_shard_group_through_ctypes(gpu):
    """mof_shard_bm_* from the ctypes binding on a one-device group, with the (1-rank) RCCL gather: per-block shifts and the
    per-pair modes come back from ONE slab ("BM mode vectors ride in the same slab"), bit-equal to the engine's
    own batch call."""
    from mrs_optic_flow_amd import _capi

    lib = _capi.load()
    B, h, w = 7, 136, 200
    cur, prev, _, _ = synth.batch_np(B, h, w, 5, k0=3)
    bm = FastSpacedBMMethod(16, 8, 8, (h, w))
    tc, tp = torch.from_numpy(cur).to(gpu), torch.from_numpy(prev).to(gpu)
    dx, dy, mode = bm.process_batch_device(tc, tp)
    blocks = dx[0].numel()
    grp = C.c_void_p()
    _capi.check(lib.mof_shard_bm_create(C.byref(bm.cfg), None, 1, C.byref(grp)))
    try:
        slab = lib.mof_shard_bm_slab_bytes(grp, B)
        assert slab % 16 == 0 and slab >= B * (2 * blocks + 8)
        out = torch.full((slab,), -1, dtype=torch.int8, device=gpu)
        pc, pp, po = (C.c_void_p * 1)(tc.data_ptr()), (C.c_void_p * 1)(tp.data_ptr()), (C.c_void_p * 1)(out.data_ptr())
        torch.cuda.synchronize()
        _capi.check(lib.mof_shard_bm_init_gather(grp))
        _capi.check(lib.mof_shard_bm_process_batch_device(grp, pc, tc.stride(0), pp, tp.stride(0), tc.stride(1), B, po, 1))
        _capi.check(lib.mof_shard_bm_sync(grp))
        got = out.cpu().numpy()
        for k in range(B):
            ox, oy, om = C.c_size_t(), C.c_size_t(), C.c_size_t()
            _capi.check(lib.mof_shard_bm_locate(grp, B, k, C.byref(ox), C.byref(oy), C.byref(om)))
            assert np.array_equal(got[ox.value:ox.value + blocks], dx[k].cpu().numpy().ravel())
            assert np.array_equal(got[oy.value:oy.value + blocks], dy[k].cpu().numpy().ravel())
            assert np.array_equal(got[om.value:om.value + 8], mode[k].cpu().numpy().ravel())
    finally:
        lib.mof_shard_bm_destroy(grp)

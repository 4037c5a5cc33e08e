"""Round-4 GPU tests: bench.py launching its own ranks, the native (C ABI) sharded batch entry, graph-lifetime advisor case,
seeded regression classes from the fuzzers."""
import gc
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import oracle_lib as O
from mrs_optic_flow_amd import FftMethod, MofError, ScaleRotationEstimator, release_captured, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 1e-4


def test_bench_launches_its_own_ranks(gpu):
    """`python bench.py --gpus 2 ...` run BARE (no RANK / WORLD_SIZE in the environment -- the shape of the driver's command): the
    parent touches no GPU, starts two fresh rank processes, relays rank 0's JSON line and exits 0. (gloo + --share-gpu: two
    ranks rehearse on the one GPU of this box; with RCCL the same code path needs N GPUs.)"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-gpu",
                        "--workload", "c2", "--batch", "16", "--steps", "3", "--warmup", "1", "--sustain-s", "0",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # only rank 0 prints, once
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["scaling"] == "weak" and line["config"]["batch_per_gpu"] == 16
    # a failing rank fails the launcher (an impossible workload argument makes argparse exit 2 in every rank)
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-gpu",
                          "--workload", "c2", "--batch", "-1", "--steps", "1", "--warmup", "0", "--sustain-s", "0",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert bad.returncode != 0


def test_release_captured_of_one_engine_leaves_other_graphs_replayable(gpu):
    """Advisor r03: release_captured(B) used to purge the process-wide parked list, freeing engine A -- closed while graph A could
    still replay -- under graph A. Two graphs; close A's engine; release B; replay A."""
    from mrs_optic_flow_amd import _capi
    from mrs_optic_flow_amd import engine as E

    lib = _capi.load()
    B, fs = 4, 256
    cur, prev, _, _ = synth.batch_np(B, fs, fs, 6, classes=False, k0=21)
    tc, tp = torch.from_numpy(cur).to(gpu), torch.from_numpy(prev).to(gpu)
    fa, fb = FftMethod(fs, 64, 80.0), FftMethod(fs, 128, 80.0)
    want_a = fa.process_batch_device(tc, tp).clone()
    torch.cuda.synchronize()
    parked0 = lib.mof_deferred_count()
    ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        with torch.cuda.graph(ga, stream=side):
            out_a = fa.process_batch_device(tc, tp)
        with torch.cuda.graph(gb, stream=side):
            out_b = fb.process_batch_device(tc, tp)
    # close engine A while graph A lives: the library parks it (Python's keep-alive set bypassed, as a C++ host would)
    E._CAPTURED.discard(fa)
    del fa
    gc.collect()
    assert lib.mof_deferred_count() == parked0 + 1
    del gb
    assert release_captured(fb) == 1              # B's graphs are gone ...
    assert lib.mof_deferred_count() == parked0 + 1  # ... which says nothing about A: still parked, not freed
    out_a.zero_()
    ga.replay()
    torch.cuda.synchronize()
    assert torch.equal(out_a, want_a)
    del ga, out_b
    release_captured()                            # every graph is gone: now the parked engines are freed
    assert lib.mof_deferred_count() == 0


def test_native_sharded_entry_from_a_cpp_host(gpu):
    """tests/cpp/test_shard.cpp: mof_shard_fft_* -- one process, one engine and stream per device, ceil(B / G) contiguous
    shards, ONE in-place RCCL all-gather (ncclCommInitAll + ncclAllGather through the run-time-bound librccl) -- with the
    devices this box has; every device's gathered result equals the single-engine result on the whole batch bit for bit."""
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
            _capi.check(lib.mof_shard_fft_process_batch_device(grp, pc, tc.stride(0), pp, tp.stride(0), tc.stride(1), B, po, gather))
            _capi.check(lib.mof_shard_fft_sync(grp))
            assert torch.equal(out, want), gather
    finally:
        lib.mof_shard_fft_destroy(grp)

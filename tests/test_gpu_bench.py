"""GPU tests of bench.py as a program: the two-rank rehearsals of the torch.distributed path (gloo, both ranks on the one GPU), bench.py
launching its own ranks, and (r06) the NATIVE shard group mode `--native` (mof_shard_*: one process, RCCL inside the library)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-4
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_bench_two_rank_rehearsal(gpu):
    """bench.py --gpus 2 as two fresh processes under a gloo rendezvous, both on the one GPU: the N > 1 code path
    (sharding, per-step gather, max-over-ranks timing, rank-0 JSON line) runs end to end."""
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo",
                                       "--share-gpu", "--workload", "c4", "--batch", "8", "--steps", "3", "--warmup", "1",
                                       "--sustain-s", "0.2", "--no-cpu-baseline"],  # (> 0: the cold-burst and sustained legs run with two ranks too)
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env))
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-1500:] for o in outs]
    line = json.loads([l for l in outs[0][0].splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["scaling"] == "weak"
    assert "gloo" in line["config"]["gather"] and line["config"]["batch_per_gpu"] == 8
    assert line["cold_burst"]["value"] > 0 and line["sustained"]["value"] > 0 and "native_shard_group" not in line
    assert not [l for l in outs[1][0].splitlines() if l.startswith("{")]  # only rank 0 prints


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


def _bench(*argv, env=None, timeout=900):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True, timeout=timeout, cwd=ROOT,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **(env or {})))
    assert r.returncode == 0, (argv, r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    # the contract's stdout is ONE JSON line: RCCL's version banner (printed to stdout when a communicator is built) must not be on it
    assert [l for l in r.stdout.splitlines() if l.strip()] == lines, r.stdout[:1500]
    return json.loads(lines[0])


def test_bench_native_shard_group_on_one_device(gpu):
    """`bench.py --native --gpus 1`: the C++ shard group timed by the bench -- one process, no torch.distributed,
    mof_shard_fft_create + init_gather (ncclCommInitAll) + one group call per step with the in-place all-gather of a ONE-rank communicator.
    The line records what RCCL formed (`rccl_ranks`, ncclCommCount), the per-device step time by HIP events on the shard's stream, that the
    gathered slab equals the plain engine's result bit for bit, and a rate within 5 % of the plain single-engine run in the same process
    regime (the 1-rank gather is a device-local copy of 1 MB per step). Same for c3 through mof_shard_bm_*."""
    plain = _bench("--workload", "c2", "--steps", "100", "--warmup", "10", "--sustain-s", "0.5", "--no-cpu-baseline", "--no-others")
    nat = _bench("--native", "--gpus", "1", "--workload", "c2", "--steps", "100", "--warmup", "10")
    assert nat["n_gpus"] == 1 and nat["rccl_ranks"] == 1 and nat["config"]["parallelism"] == "native shard group x1"
    g = nat["native_shard_group"]
    assert g["shard_results_equal_plain_engine"] is True and len(g["per_device_step_ms"]) == 1 and "ncclAllGather" in g["gather"]
    assert nat["roofline"]["frac"] > 0.05
    assert abs(nat["value"] / plain["value"] - 1.0) < 0.05, (nat["value"], plain["value"])
    bm = _bench("--native", "--gpus", "1", "--workload", "c3", "--steps", "30", "--warmup", "5")
    assert bm["rccl_ranks"] == 1 and bm["native_shard_group"]["shard_results_equal_plain_engine"] is True and bm["dtype"] == "u8"
    # the plain line reports both clock regimes and carries the native record of its own workload
    assert plain["cold_burst"]["value"] > 0 and plain["cold_burst"]["steps"] == 100 and "settled" in plain["value_regime"]


def test_bench_native_shard_group_rehearsal_with_two_shards(gpu):
    """`--native --gpus 2 --share-gpu`: two shards on the one device (MOF_SHARD_SHARE_DEVICE=1), no gather -- the G > 1 slab arithmetic
    of the group under the bench's protocol; shard 0's slab equals the plain engine's result."""
    line = _bench("--native", "--gpus", "2", "--share-gpu", "--workload", "c2", "--batch", "64", "--steps", "10", "--warmup", "2")
    g = line["native_shard_group"]
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 0 and g["gather"].startswith("none") and g["devices"] == [0, 0]
    assert g["shard_results_equal_plain_engine"] is True and len(g["per_device_step_ms"]) == 2 and "roofline" not in line

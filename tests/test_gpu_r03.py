"""GPU tests added in round 3:

* BASELINE c5's scale/rotation stage at its FULL size through the shipped configuration: 480^2, M = 49.9, the default
  1024-pair passes, 1100 pairs (a full pass + a ragged second), both lane settings, and 512-pair passes;
* engine lifetime under HIP graphs: a captured batch survives the loss of every Python reference to its engines, a
  garbage collection inside and after the capture, and a later eager batch that would have grown the scratch;
* the non-blocking RCCL gather (sharding.AsyncGather) in a fresh 1-rank child process, incl. its consumer guard;
* bench.py's N > 1 path rehearsed as two fresh gloo ranks sharing the one GPU;
* the remap kernel's OTHER forms (16-deep staging ring, one box per wave instead of per super-tile), which the shipped maps no
  longer select by themselves, byte for byte -- in child processes, because the selecting knobs are read once per process.
"""
import gc
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

import oracle_lib as O
import sr_scenes
from mrs_optic_flow_amd import FftMethod, MofError, ScaleRotationEstimator, release_captured
from mrs_optic_flow_amd.engine import INTER_CUBIC, INTER_LANCZOS4

pytestmark = pytest.mark.gpu
TOL = 1e-4  # px, north_star's bar for the FFT path (absolute)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("lanes,chunk", [(1, 0), (2, 0), (1, 1024)])
def test_c5_scale_rotation_full_size_default_passes(gpu, lanes, chunk):
    """scaleRotationEstimator.cpp:34-148 at BASELINE c5's size: 1100 pairs of 480 x 480 crops through the default
    1024-pair passes (and through 512-pair ones). Every sampled pair -- on both sides of the pass boundaries and in the ragged tail -- equals the
    same pair processed alone, bit for bit; pairs that repeat a prototype repeat its bits; six samples match the oracle
    (pt within 1e-4 px, scale / rot within 1e-5); the log-polar bytes of a sample match the oracle's byte for byte."""
    res, M, B = 480, 49.9, 1100
    base = sr_scenes.canvas(331, res)
    protos = [(1.0, 0.0), (1.03, 2.0), (0.96, -3.0), (1.0, 5.0), (1.06, -1.0), (0.92, 7.0), (1.01, 0.5), (0.99, -0.25),
              (1.02, 11.0)]
    views = np.stack([sr_scenes.view(base, res, s, r) for s, r in protos])
    P = len(protos)
    idx = np.arange(B) % P
    pidx = (idx * 4 + 1) % P
    # frames live inside 752-wide rows like the c5 crop of the camera frame (pitch 752, crop origin as the pointer)
    wide = torch.zeros((2, P, res, 752), dtype=torch.uint8, device=gpu)
    wide[:, :, :, 136:136 + res] = torch.from_numpy(views).to(gpu)
    cur = wide[0][torch.from_numpy(idx).to(gpu)][:, :, 136:136 + res]
    prev = wide[1][torch.from_numpy(pidx).to(gpu)][:, :, 136:136 + res]
    est = ScaleRotationEstimator(res, M, pipeline_lanes=lanes, batch_chunk=chunk)  # 0: the library's default pass
    got = est.process_batch_device(cur, prev)
    torch.cuda.synchronize()
    got = got.cpu().numpy()
    assert np.isfinite(got).all()
    for k in (0, 1, 255, 510, 511, 512, 513, 767, 1022, 1023, 1024, 1025, 1098, 1099):
        alone = est.process_batch_device(cur[k:k + 1], prev[k:k + 1]).cpu().numpy()[0]
        assert np.array_equal(alone, got[k]), (k, alone, got[k])
    for k in range(P, B):
        assert np.array_equal(got[k], got[k % P]), k
    for k in (0, 3, 511, 512, 1024, 1099):
        ref = O.ScaleRotationEstimator(res, M, 64)
        ref.processImage(views[pidx[k]])
        s, r = ref.processImage(views[idx[k]])
        assert np.allclose(got[k, 2:], ref.pt, rtol=0, atol=TOL), (k, got[k], ref.pt)
        assert abs(got[k, 0] - s) < 1e-5 and abs(got[k, 1] - r) < 1e-5
    # the remap stage of samples from each pass, through the same engine, byte for byte
    sample = [0, 511, 512, 1099]
    lp = est.logpolar_batch_device(cur[sample], INTER_LANCZOS4).cpu().numpy()
    lc = est.logpolar_batch_device(prev[sample], INTER_CUBIC).cpu().numpy()
    for j, k in enumerate(sample):
        assert np.array_equal(lp[j], O.logpolar(views[idx[k]], M, INTER_LANCZOS4)), k
        assert np.array_equal(lc[j], O.logpolar(views[pidx[k]], M, INTER_CUBIC)), k


def test_captured_graph_outlives_every_python_reference(gpu):
    """The cause of round 2's two GPU memory faults: a captured batch holds raw pointers into engine-owned memory, and
    the engines' finalizers freed it. Now a captured call pins its engine (library: deferred destroy, scratch that
    cannot move; Python: a keep-alive set). Capture, drop every reference, collect, replay -- once -- and compare."""
    from mrs_optic_flow_amd import _capi, synth

    lib = _capi.load()
    B, fs = 6, 256
    cur, prev, _, _ = synth.batch_np(B, fs, fs, 6, classes=False, k0=11)
    tc, tp = torch.from_numpy(cur).to(gpu), torch.from_numpy(prev).to(gpu)
    fm, sr = FftMethod(fs, 64, 80.0), ScaleRotationEstimator(fs, 45.0)
    want = fm.process_batch_device(tc, tp).clone()
    want_sr = sr.process_batch_device(tc, tp).clone()
    torch.cuda.synchronize()
    parked_before = lib.mof_deferred_count()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            # an unrelated engine created, used and finalised INSIDE the capture window (what Python's collector may do
            # at any time): the library allocates and frees under the relaxed capture mode, the capture stays valid
            tmp = FftMethod(fs, 32, 80.0)
            del tmp
            gc.collect()
            out = fm.process_batch_device(tc, tp)
            sr_out = sr.process_batch_device(tc, tp)
    assert sr.graph_pinned
    # growing the pinned scratch must be refused (it would move memory under the graph), loudly
    big_c = tc.repeat(4, 1, 1)
    with pytest.raises(MofError) as exc:
        sr.process_batch_device(big_c, big_c)
    assert exc.value.code == _capi.MOF_ERR_BUSY and "graph" in str(exc.value)
    del exc  # (its traceback holds the frame of sr.process_batch_device, and with it the engine)
    # the C ABI's own protection, without Python's keep-alive set: destroying a pinned engine parks it
    from mrs_optic_flow_amd import engine as E
    E._CAPTURED.discard(fm)
    E._CAPTURED.discard(sr)
    del fm, sr
    gc.collect()
    assert lib.mof_deferred_count() == parked_before + 2
    out.zero_()
    sr_out.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, want) and torch.equal(sr_out, want_sr)
    del g
    assert lib.mof_purge_deferred() >= 2 and lib.mof_deferred_count() == 0


def test_release_captured_lets_the_scratch_grow_again(gpu):
    res = 240
    base = sr_scenes.canvas(5, res)
    v = np.stack([sr_scenes.view(base, res, 1.0 + 0.01 * k, 1.0 * k) for k in range(5)])
    cur, prev = torch.from_numpy(v[1:]).to(gpu), torch.from_numpy(v[:-1]).to(gpu)
    est = ScaleRotationEstimator(res, 40.0)
    est.reserve(4)
    want = est.process_batch_device(cur, prev).clone()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            out = est.process_batch_device(cur, prev)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, want)
    with pytest.raises(MofError):
        est.process_batch_device(cur.repeat(8, 1, 1), prev.repeat(8, 1, 1))
    del g
    assert release_captured(est) == 1 and not est.graph_pinned
    got = est.process_batch_device(cur.repeat(8, 1, 1), prev.repeat(8, 1, 1))
    torch.cuda.synchronize()
    assert torch.equal(got[:4], want) and torch.equal(got[28:], want)


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


def test_async_gather_on_rccl_in_a_fresh_process(gpu):
    """sharding.AsyncGather over a 1-rank RCCL group (the only NCCL world a 1-GPU box allows), in a child process so
    that the test runner itself never initialises a process group."""
    script = _GATHER_SCRIPT.format(root=ROOT, port=_free_port())
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "async gather ok" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


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
                                       "--sustain-s", "0", "--no-cpu-baseline"],
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env))
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-1500:] for o in outs]
    line = json.loads([l for l in outs[0][0].splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["scaling"] == "weak"
    assert "gloo" in line["config"]["gather"] and line["config"]["batch_per_gpu"] == 8
    assert not [l for l in outs[1][0].splitlines() if l.startswith("{")]  # only rank 0 prints


_REMAP_SCRIPT = r"""
import sys
sys.path[:0] = [{root!r}, {tests!r}]
import numpy as np, torch
import oracle_lib as O, sr_scenes
from mrs_optic_flow_amd import ScaleRotationEstimator
from mrs_optic_flow_amd.engine import INTER_CUBIC, INTER_LANCZOS4
dev = torch.device("cuda", 0)
checked = 0
for res, M in ((480, 49.9), (256, 45.0)):
    base = sr_scenes.canvas(900 + res, res)
    frames = np.stack([sr_scenes.view(base, res, 1.0 + 0.02 * k, 3.0 * k - 9.0) for k in range(9)])
    frames[4, :7, :] = 255
    est = ScaleRotationEstimator(res, M)
    t = torch.from_numpy(frames).to(dev)
    for interp in (INTER_CUBIC, INTER_LANCZOS4):
        got = est.logpolar_batch_device(t, interp).cpu().numpy()
        for k in range(9):
            want = O.logpolar(frames[k], M, interp)
            assert np.array_equal(got[k], want), (res, interp, k, int((got[k] != want).sum()))
            checked += 1
print("remap ok", checked)
"""


@pytest.mark.parametrize("env", [{"MOF_SR_LP_RING": "16"}, {"MOF_SR_LP_SUPER": "0"}, {"MOF_SR_LP_STAGED": "0"}])
def test_remap_other_kernel_forms_are_byte_exact(gpu, env):
    """K4's 16-deep ring (maps whose largest super-tile box exceeds 3072 dwords), its one-box-per-wave form (resolutions that
    are not a multiple of 16, boxes beyond 4096 dwords) and the table-in-LDS kernel (unaligned layouts), forced by their
    knobs on maps that would take the 12-deep super-tile form: every byte against the oracle."""
    script = _REMAP_SCRIPT.format(root=ROOT, tests=os.path.join(ROOT, "tests"))
    r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
    assert r.returncode == 0 and "remap ok 36" in r.stdout, (env, r.stdout[-1500:], r.stderr[-1500:])


@pytest.mark.parametrize("env,target,select", [
    ({"MOF_SR_PAIR_SEQ": "0"}, "tests/test_gpu_sr.py", "batch_pairs or black_frames or opencv3"),
    ({"MOF_FFT_SEQ_HALF64": "1"}, "tests/test_gpu_fft_sequence.py", "video_matches or known_answers or wider_allocation"),
])
def test_knob_selected_kernel_forms_pass_their_parity_tests(gpu, env, target, select):
    """The packed pair kernels of the estimator (K5 / K6, `MOF_SR_PAIR_SEQ=0`) and the half-tile form of the 64 x 64 sequence
    kernel (`MOF_FFT_SEQ_HALF64=1`) stay in the library as A/B forms: the parity tests of the shipped forms, run once more in a
    child process with the knob set (the knobs are read once per process)."""
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, target), "-m", "gpu", "-x", "-q", "-k", select,
                        "-p", "no:cacheprovider"], capture_output=True, text=True, timeout=900, cwd=ROOT,
                       env=dict(os.environ, **env))
    assert r.returncode == 0 and " passed" in r.stdout, (env, r.stdout[-2000:], r.stderr[-1000:])

"""GPU tests of HIP-graph capture of the batch entries: multi-pass batches, the lifetime of captured engines and scratch
(release_captured), several engines' graphs side by side."""
import gc
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import sr_scenes
from mrs_optic_flow_amd import FftMethod, MofError, ScaleRotationEstimator, release_captured, synth

pytestmark = pytest.mark.gpu
TOL = 1e-4
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_multi_pass_batch_is_graph_capturable(gpu):
    """A scale/rotation batch longer than one pipeline pass, on one lane and as two lanes (pipeline_lanes = 2: remaps on
    the engine's own stream beside the transforms). Under HIP-graph capture the engine's stream joins the caller's capture
    by an event fork / join and the graph must replay to the bits of the eager run. Run in a child process (batch_chunk =
    2: seven pairs = four passes) at 240 and 480."""
    script = os.path.join(ROOT, "tools", "check_graph_capture.py")
    for res in ("240", "480"):
        r = subprocess.run([sys.executable, script, res], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and f"graph ok {res}" in r.stdout, (r.stdout + r.stderr)[-2000:]


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

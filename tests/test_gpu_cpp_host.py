"""GPU test of the C++ host mirror (include/mof/processors.hpp) through tests/cpp/test_processors:
the stateful call sequence the ROS node makes (setImPrev(zeros), then processImage per frame)."""
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O
from mrs_optic_flow_amd import synth

pytestmark = pytest.mark.gpu
BIN = os.path.join(os.path.dirname(__file__), "cpp", "test_processors")


def _run(args, frames, tmp_path):
    assert os.path.exists(BIN), "tests/cpp/test_processors missing: run __graft_entry__.build()"
    path = tmp_path / "frames.raw"
    frames.tofile(path)
    out = subprocess.run([BIN] + [str(a) for a in args] + [str(path)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    return [line.split() for line in out.stdout.strip().splitlines()]


def test_fft_method_sequence(gpu, tmp_path):
    fs, sps, n = 192, 64, 4
    frames = np.stack([synth.pair_np(33, fs, fs, 2 * t, -3 * t)[0] for t in range(n)])
    lines = _run(["fft", fs, sps, 80, n], frames, tmp_path)
    lay = O.fft_layout(fs, fs, sps, 3, 3)
    for t, tok in enumerate(lines):
        assert tok[0] == "frame" and int(tok[1]) == t and int(tok[3]) == 9
        got = np.array([float(v) for v in tok[4:]]).reshape(9, 2)
        prev = frames[t] if t == 0 else frames[t - 1]  # `first`: the first frame correlates with itself
        want, _ = O.fft_process(frames[t], prev, lay, 64)
        assert np.allclose(got, want, rtol=0, atol=1e-4, equal_nan=True)


def test_fft_method_ocl_peak_model_sequence(gpu, tmp_path):
    """The constructor's trailing peak_model argument selects the OpenCL kernel's peak model (MOF_PEAK_OCL)."""
    fs, sps, n = 192, 64, 3
    frames = np.stack([synth.pair_np(35, fs, fs, 3 * t, -2 * t)[0] for t in range(n)])
    lines = _run(["fftocl", fs, sps, 80, n], frames, tmp_path)
    lay = O.fft_layout(fs, fs, sps, 3, 3)
    for t, tok in enumerate(lines):
        got = np.array([float(v) for v in tok[4:]]).reshape(9, 2)
        prev = frames[t] if t == 0 else frames[t - 1]
        want, _ = O.fft_process_ocl(frames[t], prev, lay, 55, 64)
        assert np.allclose(got, want, rtol=0, atol=1e-4, equal_nan=True)


def test_block_method_sequence(gpu, tmp_path):
    fs, sps, r, n = 144, 32, 8, 3
    frames = np.stack([synth.pair_np(34, fs, fs, 3 * t, t, blur=False)[0] for t in range(n)])
    lines = _run(["bm", fs, sps, r, n], frames, tmp_path)
    cfg = O.bm_config_block_method(fs, sps, r)
    prev = np.zeros((fs, fs), np.uint8)  # BlockMethod.cpp:17-18
    for t, tok in enumerate(lines):
        dx, dy, mode = O.bm_process(frames[t], prev, cfg)
        assert (int(tok[3]), int(tok[4])) == mode
        # BlockMethod::processImage returns Refine(mode, 2) (BlockMethod.cpp:79), reproduced literally by default
        assert (float(tok[6]), float(tok[7])) == O.bm_refine(frames[t], prev, mode, 2, True)
        got = np.array([int(v) for v in tok[9:]]).reshape(-1, 2)
        assert (got[:, 0] == dx.ravel()).all() and (got[:, 1] == dy.ravel()).all()
        prev = frames[t]


def test_fast_spaced_bm_sequence(gpu, tmp_path):
    w, h, sps, step, r, n = 168, 120, 16, 8, 10, 3
    frames = np.stack([synth.pair_np(35, h, w, -2 * t, 3 * t)[0] for t in range(n)])
    lines = _run(["fsbm", w, h, sps, step, r, n], frames, tmp_path)
    cfg = O.bm_config_fast_spaced(w, h, sps, step, r)
    prev = np.zeros((h, w), np.uint8)
    for t, tok in enumerate(lines):
        dx, dy, mode = O.bm_process(frames[t], prev, cfg)
        assert (float(tok[3]), float(tok[4])) == mode
        got = np.array([int(v) for v in tok[6:]]).reshape(-1, 2)
        assert (got[:, 0] == dx.ravel()).all() and (got[:, 1] == dy.ravel()).all()
        prev = frames[t]


def test_drop_in_adapter_against_the_reference_interface_header(gpu, tmp_path):
    """tests/cpp/test_adapter.cpp: `MofFftMethod : OpticFlowCalc` compiled against /root/reference/include/OpticFlowCalc.h
    (with type-check stand-ins for the OpenCV headers, tests/stubs/) and driven exactly like the node drives its
    FftMethod (optic_flow.cpp:1001-1002, :1016-1018, :1685-1690): setImPrev(zeros), processImage through the abstract
    interface, processImageLongRange on the concrete type. The binary is built where the reference exists (the build
    container) and travels with the snapshot."""
    binary = os.path.join(os.path.dirname(__file__), "cpp", "test_adapter")
    assert os.path.exists(binary), "tests/cpp/test_adapter missing: run __graft_entry__.build() where /root/reference exists"
    fs, sps, n = 256, 64, 5   # sqNum 4 -> long-range grid 1 x 1
    frames = np.stack([synth.pair_np(37, fs, fs, 4 * t, -2 * t)[0] for t in range(n)])
    path = tmp_path / "frames.raw"
    frames.tofile(path)
    out = subprocess.run([binary, str(fs), str(sps), "80", str(n), str(path)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr + out.stdout
    lines = [line.split() for line in out.stdout.strip().splitlines()]
    assert lines[-1] == ["wrong-size", "refused"]
    lay = O.fft_layout(fs, fs, sps, 4, 4)
    for t, tok in enumerate(lines[:-1]):
        assert tok[0] == "frame" and int(tok[1]) == t
        prev = frames[t] if t == 0 else frames[t - 1]  # `first` (FftMethod.cpp:1791-1793); setImPrev(zeros) does not clear it
        if tok[2] == "lr":
            assert t % 2 == 1 and int(tok[4]) == 1
            want, _ = O.fft_process_long_range(frames[t], prev, lay, 64)
        else:
            assert int(tok[4]) == 16
            want, _ = O.fft_process(frames[t], prev, lay, 64)
        got = np.array([float(v) for v in tok[5:]]).reshape(-1, 2)
        assert np.allclose(got, want, rtol=0, atol=1e-4, equal_nan=True), (t, tok[2])


def test_cpp_mirror_sequence_entries(gpu, tmp_path):
    """mof::FftMethod::processSequenceDevice and mof::scaleRotationEstimator::processSequenceDevice (a video on the device
    through the C++ mirror): K1's sequence kernel against the oracle pair by pair, the estimator's sequence entry equal to
    its own stateful loop bit for bit and to the oracle within the usual bars."""
    import sr_scenes

    fs, sps, n = 192, 64, 6
    frames = np.stack([synth.pair_np(33, fs, fs, 2 * t, -3 * t)[0] for t in range(n)])
    lines = _run(["fftseq", fs, sps, 80, n], frames, tmp_path)
    lay = O.fft_layout(fs, fs, sps, 3, 3)
    assert len(lines) == n - 1
    for t, tok in enumerate(lines):
        assert tok[0] == "pair" and int(tok[1]) == t and int(tok[3]) == 9
        got = np.array([float(v) for v in tok[4:]]).reshape(9, 2)
        want, _ = O.fft_process(frames[t + 1], frames[t], lay, 64)
        assert np.allclose(got, want, rtol=0, atol=1e-4, equal_nan=True)
    # the same video from HOST memory (FftMethod::processVideo -> mof_fft_process_batch_host): what the stateful processImage calls return
    stateful = _run(["fft", fs, sps, 80, n], frames, tmp_path)
    lines = _run(["fftvideo", fs, sps, 80, n], frames, tmp_path)
    assert len(lines) == n - 1
    for t, tok in enumerate(lines):
        want = np.array([float(v) for v in stateful[t + 1][4:]]).reshape(9, 2)
        got = np.array([float(v) for v in tok[4:]]).reshape(9, 2)
        assert np.array_equal(got, want, equal_nan=True)  # (N = 64: the pair kernel behind both)
    res, M, nf = 240, 40.0, 5
    base = sr_scenes.canvas(8, res)
    video = np.stack([sr_scenes.view(base, res, 1.0 + 0.012 * t, 1.4 * t) for t in range(nf)])
    lines = _run(["srseq", res, M, nf], video, tmp_path)
    assert lines[0] == ["gated", "0"]
    ref = O.ScaleRotationEstimator(res, M, 64)
    for t, tok in enumerate(lines[1:]):
        seq = [float(v) for v in tok[3:7]]
        stateful = [float(v) for v in tok[8:10]]
        assert seq[:2] == stateful  # the same kernels behind both entries
        ws, wr = ref.processImage(video[t])
        assert abs(seq[0] - ws) < 1e-5 and abs(seq[1] - wr) < 1e-5
        if t > 0:
            assert np.allclose(seq[2:], ref.pt, rtol=0, atol=1e-4)

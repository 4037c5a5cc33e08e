"""GPU tests of the scale/rotation estimator's SEQUENCE mode (mof_sr_process_sequence_device, csrc/sr_seq_kernel.hip):
the reference's steady state (scaleRotationEstimator.cpp:34-148) on a video that lives on the device -- first frame
INTER_CUBIC, every later frame INTER_LANCZOS4 once, frame k correlated with frame k-1, the gate of :119-121.

Bars: bit-identical to the stateful mof_sr_process fed the same frames one at a time (the same kernels run); against the
oracle (a loop over oracle ScaleRotationEstimator.processImage) pt within 1e-4 px, scale / rot within 1e-5."""
import numpy as np
import pytest
import torch

import oracle_lib as O
import sr_scenes
from mrs_optic_flow_amd import ScaleRotationEstimator

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _video(seed, res, n):
    base = sr_scenes.canvas(seed, res)
    return np.stack([sr_scenes.view(base, res, 1.0 + 0.012 * np.sin(0.7 * t) + 0.004 * t, 1.3 * t - 0.05 * t * t) for t in range(n)])


@pytest.mark.parametrize("res,M,n,chunk", [(240, 40.0, 23, 5), (256, 45.0, 9, 0), (480, 49.9, 12, 4)])
def test_sequence_equals_frame_by_frame_and_oracle(gpu, res, M, n, chunk):
    frames = _video(7 + res, res, n)
    # the video sits inside a wider buffer: pitch > res, crop origin passed as the pointer
    wide = torch.zeros((n, res + 2, res + 24), dtype=torch.uint8, device=gpu)
    wide[:, 1:1 + res, 8:8 + res] = torch.from_numpy(frames).to(gpu)
    video = wide[:, 1:1 + res, 8:8 + res]
    seq = ScaleRotationEstimator(res, M, batch_chunk=chunk)  # chunk 5 / 4: the sequence crosses several pipeline passes
    got = seq.process_sequence_device(video)
    assert seq.last_gated == 0
    got = got.cpu().numpy()
    one, ref = ScaleRotationEstimator(res, M), O.ScaleRotationEstimator(res, M, 64)
    for k in range(n):
        s, r = one.processImage(frames[k])
        assert (s, r) == (got[k, 0], got[k, 1]), (k, s, r, got[k])  # the same kernels: the same bits
        ws, wr = ref.processImage(frames[k])
        assert abs(got[k, 0] - ws) < 1e-5 and abs(got[k, 1] - wr) < 1e-5, (k, got[k], ws, wr)
        if k > 0:
            assert np.allclose(got[k, 2:], ref.pt, rtol=0, atol=TOL), (k, got[k], ref.pt)
    assert tuple(got[0]) == (1.0, 0.0, 0.0, 0.0)  # :74
    # the asynchronous form (no gate resolution) gives the same numbers when nothing is gated
    again = ScaleRotationEstimator(res, M, batch_chunk=chunk).process_sequence_device(video, resolve_gate=False)
    torch.cuda.synchronize()
    assert np.array_equal(again.cpu().numpy(), got)


def test_sequence_continues_the_stateful_sequence(gpu):
    """Chunks of a video through one engine == the whole video in one call == frames mixed with stateful calls."""
    res, M, n = 240, 40.0, 17
    frames = _video(3, res, n)
    video = torch.from_numpy(frames).to(gpu)
    whole = ScaleRotationEstimator(res, M).process_sequence_device(video).cpu().numpy()
    est = ScaleRotationEstimator(res, M, batch_chunk=3)
    parts = [est.process_sequence_device(video[0:1]), est.process_sequence_device(video[1:8])]
    s, r = est.processImage(frames[8])  # a stateful call in between continues the same sequence
    parts.append(est.process_sequence_device(video[9:]))
    got = torch.cat(parts).cpu().numpy()
    assert np.array_equal(got[:8], whole[:8]) and np.array_equal(got[8:], whole[9:])
    assert (s, r) == (whole[8, 0], whole[8, 1])
    est.reset()  # re-arms `first`: the next frame goes through INTER_CUBIC again and returns (1, 0)
    again = est.process_sequence_device(video[:5]).cpu().numpy()
    assert np.array_equal(again, whole[:5])


def test_sequence_gate_leaves_prev_unchanged(gpu):
    """scaleRotationEstimator.cpp:119-121: a frame whose |pt.x| > res/2 returns (1, 0) and does not become `prev`.
    Unrelated noise frames gate about once in a hundred pairs; 600 of them hold gated frames (asserted), and the
    resolving sequence call must then equal the stateful frame-by-frame loop, which applies the rule on the host."""
    res, M, n = 240, 40.0, 600
    rng = np.random.default_rng(2024)
    frames = rng.integers(0, 256, (n, res, res), dtype=np.uint8)
    video = torch.from_numpy(frames).to(gpu)
    raw = ScaleRotationEstimator(res, M, batch_chunk=128).process_sequence_device(video, resolve_gate=False)
    torch.cuda.synchronize()
    raw = raw.cpu().numpy()
    raw_gated = np.flatnonzero(np.abs(raw[:, 2]) > res / 2)
    assert raw_gated.size > 0, "no gated frame among 600 noise frames: extend the video"
    assert np.all(raw[raw_gated, 0] == 1.0) and np.all(raw[raw_gated, 1] == 0.0)
    est = ScaleRotationEstimator(res, M, batch_chunk=128)
    got = est.process_sequence_device(video).cpu().numpy()
    one = ScaleRotationEstimator(res, M)
    want = np.array([one.processImage(f) for f in frames])
    assert np.array_equal(got[:, :2], want)
    gated = np.flatnonzero(np.abs(got[:, 2]) > res / 2)
    assert est.last_gated == gated.size and gated.size >= 1
    # frames up to and including the first gated one cannot differ from the unresolved run; the frame behind it was
    # correlated with an OLDER partner and does
    g0 = int(gated[0])
    assert np.array_equal(got[:g0 + 1], raw[:g0 + 1])
    if g0 + 1 < n:
        assert not np.array_equal(got[g0 + 1], raw[g0 + 1])
    # both engines were left in the same state: the next frame gives the same answer
    nxt = rng.integers(0, 256, (res, res), dtype=np.uint8)
    assert est.processImage(nxt) == one.processImage(nxt)


def test_sequence_full_size_properties(gpu):
    """c5seq at BASELINE's size: 480^2, M = 49.9, default passes, 1100 frames (two full passes and a ragged third). The video
    cycles through nine views, so the pair (k-1, k) repeats with period nine: equal pairs give equal bits wherever they
    sit; the head equals the frame-by-frame engine; samples match the oracle."""
    res, M, n = 480, 49.9, 1100
    base = sr_scenes.canvas(77, res)
    protos = np.stack([sr_scenes.view(base, res, 1.0 + 0.01 * ((3 * t) % 9 - 4), 0.8 * ((5 * t) % 9 - 4)) for t in range(9)])
    idx = np.arange(n) % 9
    video = torch.from_numpy(protos).to(gpu)[torch.from_numpy(idx).to(gpu)]
    est = ScaleRotationEstimator(res, M)
    got = est.process_sequence_device(video).cpu().numpy()
    assert est.last_gated == 0 and np.isfinite(got).all()
    for k in range(11, n):  # (pair (0, 1) has the CUBIC first frame: the period starts at pair (1, 2))
        assert np.array_equal(got[k], got[k - 9]), k
    one = ScaleRotationEstimator(res, M)
    for k in range(12):
        assert one.processImage(protos[idx[k]]) == (got[k, 0], got[k, 1]), k
    ref = O.ScaleRotationEstimator(res, M, 64)
    for k in range(11):
        ws, wr = ref.processImage(protos[idx[k]])
        assert abs(got[k, 0] - ws) < 1e-5 and abs(got[k, 1] - wr) < 1e-5
        if k > 0:
            assert np.allclose(got[k, 2:], ref.pt, rtol=0, atol=TOL)

"""GPU parity tests of the scale/rotation estimator (K4..K8) through the C ABI against oracle/lp_ref.c:
the log-polar remap is integer work (bit-exact), the whole-frame phase correlation is held to 1e-4 px on pt."""
import numpy as np
import pytest
import torch

import oracle_lib as O
import sr_scenes
from mrs_optic_flow_amd import ScaleRotationEstimator

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("res,M", [(240, 40.0), (256, 45.0), (480, 49.9)])
def test_batch_pairs_match_oracle(gpu, res, M):
    base = sr_scenes.canvas(3 + res, res)
    params = [(1.0, 0.0), (1.03, 2.0), (0.96, -3.0), (1.0, 5.0)]
    frames = np.stack([sr_scenes.view(base, res, s, r) for s, r in params])
    prev = np.repeat(frames[:1], len(params), 0)
    # frames embedded in a wider buffer: pitch > res, the crop origin is passed as the pointer
    big = torch.zeros((2, len(params), res + 3, res + 40), dtype=torch.uint8, device=gpu)
    big[0, :, 1:1 + res, 24:24 + res] = torch.from_numpy(frames).to(gpu)
    big[1, :, 1:1 + res, 24:24 + res] = torch.from_numpy(prev).to(gpu)
    est = ScaleRotationEstimator(res, M)
    got = est.process_batch_device(big[0, :, 1:1 + res, 24:24 + res], big[1, :, 1:1 + res, 24:24 + res]).cpu().numpy()
    for k in range(len(params)):
        ref = O.ScaleRotationEstimator(res, M, 64)
        ref.processImage(prev[k])
        s, r = ref.processImage(frames[k])
        assert np.allclose(got[k, 2:], ref.pt, rtol=0, atol=1e-4), (k, got[k], ref.pt)
        assert abs(got[k, 0] - s) < 1e-5 and abs(got[k, 1] - r) < 1e-5
    # identical frames: prev went through INTER_CUBIC, cur through INTER_LANCZOS4 (:45 vs :112) -> nearly, not exactly, (1, 0)
    assert abs(got[0, 0] - 1.0) < 2e-3 and abs(got[0, 1]) < 2e-3


def test_stateful_sequence_and_gate(gpu):
    res, M = 240, 40.0
    base = sr_scenes.canvas(17, res)
    seq = [sr_scenes.view(base, res, 1.0 + 0.02 * t, 1.5 * t) for t in range(4)]
    est, ref = ScaleRotationEstimator(res, M), O.ScaleRotationEstimator(res, M, 64)
    for f in seq:
        s, r = est.processImage(f)
        ws, wr = ref.processImage(f)
        assert abs(s - ws) < 1e-5 and abs(r - wr) < 1e-5
    est.reset()
    assert est.processImage(seq[2]) == (1.0, 0.0)
    # wide (non-contiguous) cv::Mat-like view
    wide = np.zeros((res, res + 32), np.uint8)
    wide[:, 8:8 + res] = seq[3]
    ref2 = O.ScaleRotationEstimator(res, M, 64)
    ref2.processImage(seq[2])
    ws, wr = ref2.processImage(seq[3])
    s, r = est.processImage(wide[:, 8:8 + res])
    assert abs(s - ws) < 1e-5 and abs(r - wr) < 1e-5


def test_opencv3_logpolar_generation(gpu):
    """logpolar_variant = LOGPOLAR_CV3: the estimator as it runs under ROS Melodic (cvLogPolar, scaleRotationEstimator.cpp:41-46)."""
    from mrs_optic_flow_amd.engine import LOGPOLAR_CV3
    res, M = 240, 40.0
    base = sr_scenes.canvas(23, res)
    seq = [sr_scenes.view(base, res, 1.0 + 0.015 * t, -1.2 * t) for t in range(4)]
    est, ref = ScaleRotationEstimator(res, M, logpolar_variant=LOGPOLAR_CV3), O.ScaleRotationEstimator(res, M, 64, variant=1)
    other = O.ScaleRotationEstimator(res, M, 64, variant=0)
    differs = False
    for f in seq:
        s, r = est.processImage(f)
        ws, wr = ref.processImage(f)
        os_, or_ = other.processImage(f)
        assert abs(s - ws) < 1e-5 and abs(r - wr) < 1e-5
        differs = differs or abs(s - os_) > 1e-7 or abs(r - or_) > 1e-7
    assert differs   # the two generations are different estimators
    cur = torch.from_numpy(np.stack(seq[1:])).to(gpu)
    prev = torch.from_numpy(np.stack(seq[:-1])).to(gpu)
    got = est.process_batch_device(cur, prev).cpu().numpy()
    for k in range(3):
        fresh = O.ScaleRotationEstimator(res, M, 64, variant=1)
        fresh.processImage(seq[k])
        fresh.processImage(seq[k + 1])
        assert np.allclose(got[k, 2:], fresh.pt, rtol=0, atol=1e-4)


def test_black_frames_give_the_reference_degenerate_answer(gpu):
    """An all-zero frame has an all-zero log-polar image: in the reference's separate transforms its spectrum is exactly zero,
    the surface is flat zero and pt = (res/2, res/2) -- scale = exp(res / 2M), rot = pi: nonsense, but deterministic, and not
    gated (|pt.x| > res/2 is false). The pair pipeline packs cur + i prev into one transform and would leak rounding noise
    into those zeros; K6 sees the exact DC bin (a sum of zero) and tells K8. The sequence / stateful kernels transform frames
    separately and need nothing."""
    res, M = 240, 40.0
    base = sr_scenes.canvas(3, res)
    v = sr_scenes.view(base, res, 1.02, 3.0)
    black = np.zeros((res, res), np.uint8)
    pairs = [(v, black), (black, v), (black, black), (v, v)]
    cur = torch.from_numpy(np.stack([a for a, _ in pairs])).to(gpu)
    prev = torch.from_numpy(np.stack([b for _, b in pairs])).to(gpu)
    est = ScaleRotationEstimator(res, M)
    got = est.process_batch_device(cur, prev).cpu().numpy()
    for k, (a, b) in enumerate(pairs):
        ref = O.ScaleRotationEstimator(res, M, 64)
        ref.processImage(b)
        s, r = ref.processImage(a)
        assert np.allclose(got[k, 2:], ref.pt, rtol=0, atol=1e-4), (k, got[k], ref.pt)
        assert abs(got[k, 0] - s) < 1e-5 * max(1.0, s) and abs(got[k, 1] - r) < 1e-5, (k, got[k], s, r)
    assert np.allclose(got[0, 2:], res / 2) and np.allclose(got[2, 2:], res / 2)
    # a black frame inside a stream: sequence entry == stateful calls == oracle
    video = np.stack([v, black, v, sr_scenes.view(base, res, 1.0, 0.0)])
    seq = ScaleRotationEstimator(res, M).process_sequence_device(torch.from_numpy(video).to(gpu)).cpu().numpy()
    one, ref = ScaleRotationEstimator(res, M), O.ScaleRotationEstimator(res, M, 64)
    for t in range(len(video)):
        s1, r1 = one.processImage(video[t])
        ws, wr = ref.processImage(video[t])
        assert (s1, r1) == (seq[t, 0], seq[t, 1])
        assert abs(seq[t, 0] - ws) < 1e-5 * max(1.0, ws) and abs(seq[t, 1] - wr) < 1e-5, (t, seq[t], ws, wr)

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

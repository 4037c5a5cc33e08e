"""CPU tests of the scale/rotation restatement (oracle/lp_ref.c): remap table invariants, log-polar geometry,
estimator state machine. Parity with OpenCV itself is unpinned (see lp_ref.c)."""
import numpy as np
import pytest

import oracle_lib as O
import sr_scenes


def test_logpolar_identity_properties():
    res, M = 240, 40.0
    flat = np.full((res, res), 93, np.uint8)
    lp = O.logpolar(flat, M, 4)
    # weights sum to exactly 2^15 -> a flat image maps to the same grey wherever the anchor is inside
    assert set(np.unique(lp)) <= {0, 93}
    # columns beyond rho = M * ln(1 + distance to the farthest corner) never land inside the image
    rho_max = int(np.ceil(M * np.log(1 + np.hypot(res / 2, res / 2))))
    assert (lp[:, rho_max + 1:] == 0).all() and (lp[:, :int(M * np.log(res / 2))] == 93).all()
    # transparent border: outliers keep the destination's previous content
    dst = np.full((res, res), 7, np.uint8)
    O.logpolar(flat, M, 2, dst)
    assert set(np.unique(dst)) <= {7, 93} and (dst[:, rho_max + 1:] == 7).all()


@pytest.mark.parametrize("interp", [2, 4])
def test_logpolar_geometry(interp):
    """Row phi, column rho samples the source at centre + (exp(rho/M) - 1)(cos, sin)(2 pi phi / res)."""
    res, M = 240, 45.0
    yy, xx = np.mgrid[0:res, 0:res]
    ramp = np.clip(xx * 0.5 + yy * 0.25 + 20, 0, 255).astype(np.uint8)  # smooth: interpolation error < 1 LSB
    lp = O.logpolar(ramp, M, interp).astype(np.float64)
    for phi, rho in [(0, 100), (60, 150), (120, 170), (200, 120)]:
        r = np.exp(rho / M) - 1
        x = r * np.cos(2 * np.pi * phi / res) + res // 2
        y = r * np.sin(2 * np.pi * phi / res) + res // 2
        assert 2 <= x < res - 3 and 2 <= y < res - 3
        assert abs(lp[phi, rho] - (x * 0.5 + y * 0.25 + 20)) <= 1.0


def test_estimator_state_machine_and_response():
    res, M = 240, 40.0
    base = sr_scenes.canvas(9, res)
    f0 = sr_scenes.view(base, res, 1.0, 0.0)
    f1 = sr_scenes.view(base, res, 1.04, 2.5)
    est = O.ScaleRotationEstimator(res, M, 64)
    assert est.processImage(f0) == (1.0, 0.0)            # first call: (1, 0), scaleRotationEstimator.cpp:74
    s, r = est.processImage(f1)
    px, py = est.pt
    assert s == pytest.approx(np.exp(px / M)) and r == pytest.approx(py / (res / 360) * np.pi / 180)
    assert 0.5 < abs(px) < 3.0 and 0.5 < abs(py) < 3.0   # ~ M ln(1.04) = 1.6 px, 2.5 deg * res/360 = 1.7 px
    s2, r2 = est.processImage(f1)                        # same frame again: previous was updated (:128)
    assert abs(s2 - 1.0) < 1e-6 and abs(r2) < 1e-6
    e32 = O.ScaleRotationEstimator(res, M, 32)
    e32.processImage(f0)
    s32, r32 = e32.processImage(f1)
    assert abs(s32 - s) < 1e-5 and abs(r32 - r) < 1e-5

"""CPU tests of the scale/rotation restatement (oracle/lp_ref.c): remap table invariants, log-polar geometry,
estimator state machine. Parity with OpenCV itself is unpinned (see lp_ref.c)."""
import numpy as np
import pytest

import oracle_lib as O
import sr_scenes
from mrs_optic_flow_amd import synth


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


def test_logpolar_generations_differ_as_documented():
    """Variant 0 = cv::logPolar of OpenCV 4.x (warpPolar form: radius exp(rho * Kmag) - 1, so rho = 0 is the centre),
    variant 1 = cvLogPolar of OpenCV 3.2 (radius exp(rho / M), rho = 0 is the unit circle) -- the two calls
    scaleRotationEstimator.cpp:41-46 / :107-113 compile under ROS Noetic / Melodic."""
    res, M = 240, 40.0
    x4, y4 = O.logpolar_maps(res, M, 0)
    x3, y3 = O.logpolar_maps(res, M, 1)
    c = res // 2
    assert np.all(x4[:, 0] == c) and np.all(y4[:, 0] == c)                      # radius 0 at rho = 0
    assert np.allclose(np.hypot(x3[:, 0] - c, y3[:, 0] - c), 1.0, atol=1e-5)    # radius 1 at rho = 0
    r4 = np.hypot(x4[0] - c, y4[0] - c)
    r3 = np.hypot(x3[0] - c, y3[0] - c)
    assert np.allclose(r3 - r4, 1.0, atol=2e-4 * r3.max())                      # the "- 1" and nothing else
    assert np.allclose(r4, np.exp(np.arange(res) / M) - 1, rtol=1e-6, atol=1e-5)
    # rows are angles over the full circle in both
    k = res // 4
    assert abs(x4[k, 50] - c) < 1e-3 and y4[k, 50] > c and abs(x3[k, 50] - c) < 1e-3
    # the remapped images differ (slightly) and both are deterministic
    src = synth.canvas_np(3, res, res, True)[:res, :res].copy()
    a, b = O.logpolar(src, M, 4, variant=0), O.logpolar(src, M, 4, variant=1)
    assert np.array_equal(a, O.logpolar(src, M, 4, variant=0)) and (a != b).mean() > 0.05
    # an estimator of either generation still recovers a planted rotation
    for variant in (0, 1):
        base = sr_scenes.canvas(5, res)
        est = O.ScaleRotationEstimator(res, M, 64, variant=variant)
        est.processImage(sr_scenes.view(base, res, 1.0, 0.0))
        s, r = est.processImage(sr_scenes.view(base, res, 1.0, 4.0))
        assert abs(np.rad2deg(r) - 4.0) < 0.6 or abs(np.rad2deg(r) + 4.0) < 0.6, (variant, s, r)

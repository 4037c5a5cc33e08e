"""Geometry tail (SURVEY 8(f) N1 getRT, N3 get2DT): the library's host forms against the oracle restatement
(oracle/geom_ref.c) and against analytic known answers. No GPU needed: the host forms are plain fp64 host code.

Bars: get2DT, undistortion and the homography decomposition are closed-form (+, -, *, /, sqrt and libm calls evaluated
in the same order by both sides on one host) -> EXACT equality with the oracle. The RANSAC stage uses the project's
own documented sampler (cv::RNG is not restated, see include/mof.h): the inlier mask must equal the oracle's and the
planted consensus set, the refined homography must agree to 1e-9.
"""
import ctypes as C

import numpy as np
import pytest

import geom_scenes as S
import oracle_lib as O
from mrs_optic_flow_amd import geometry as G

CAM = (340.0, 338.5, 376.0, 240.0, -0.28, 0.07, 0.0004, -0.0003, -0.006)  # fx, fy, cx, cy, k1, k2, p1, p2, k3


def cam_pair():
    return G.Camera(*CAM), O.GeomCamera(*CAM)


def layouts(gx, gy, origin, stride, patch):
    return (G.Layout(gx, gy, origin[0], origin[1], stride[0], stride[1], patch),
            O.GeomLayout(gx, gy, origin[0], origin[1], stride[0], stride[1], patch))


def rt_params(height, dt, ulx, ang_q, c2b_q=(0, 0, 0, 1), c2b_t=(0, 0, 0)):
    a4, c4, c3 = (C.c_double * 4)(*ang_q), (C.c_double * 4)(*c2b_q), (C.c_double * 3)(*c2b_t)
    return G.RtParams(height, dt, ulx, a4, c4, c3), O.GeomRtParams(height, dt, ulx, a4, c4, c3)


# ---- get2DT -----------------------------------------------------------------------------------------------------------

def test_get_2dt_is_exact_and_matches_the_closed_form():
    rng = np.random.default_rng(3)
    gcam, ocam = cam_pair()
    for trial in range(200):
        gx = int(rng.integers(1, 4))
        gl, ol = layouts(gx, gx, (0, 0), (120, 120), 120)
        shifts = rng.normal(0, 6, (gx * gx, 2))
        if trial % 3 == 0 and gx > 1:
            shifts[0] = np.nan           # the first VALID vector is used (optic_flow.cpp:406-409, :471)
        if trial % 7 == 0:
            shifts[: gx * gx // 2 + 1, 1] = np.inf
        vals = (float(rng.uniform(0.5, 9)), float(rng.uniform(0.002, 0.1)), float(rng.normal(0, 0.4)),
                float(rng.normal(0, 0.4)), float(rng.uniform(-3.2, 3.2)))
        st, tran, diff = G.get_2dt(shifts, gl, gcam, G.T2dParams(*vals))
        wst, wtran, wdiff = O.geom_get_2dt(shifts, ol, ocam, O.Geom2dtParams(*vals))
        assert st == wst
        assert np.array_equal(tran, wtran) and np.array_equal(diff, wdiff), (trial, tran, wtran)
        if st == 0:
            k = int(np.flatnonzero(np.isfinite(shifts).all(axis=1))[0])
            h, dt, rr, pr, yaw = vals
            xc, yc = -np.tan(rr * dt) * CAM[0] / 4, np.tan(pr * dt) * CAM[1] / 4
            tc, yw = np.hypot(xc, yc), np.arctan2(yc, xc) + yaw
            corr = np.array([np.cos(yw) * tc, np.sin(yw) * tc])
            want = -(shifts[k] + corr) * np.array([h / CAM[0] * 4, h / CAM[1] * 4]) / dt
            assert np.allclose(tran[:2], want, rtol=1e-9, atol=1e-9) and tran[2] == 0
            want_diff = -corr * np.array([h / CAM[0] * 4, h / CAM[1] * 4]) / dt
            assert np.allclose(diff[:2], want_diff, rtol=1e-7, atol=1e-9)


def test_get_2dt_early_returns():
    gcam, ocam = cam_pair()
    gl, ol = layouts(2, 2, (0, 0), (120, 120), 120)
    nan = np.full((4, 2), np.nan)
    good = np.ones((4, 2))
    assert G.get_2dt(nan, gl, gcam, G.T2dParams(2, 0.02, 0, 0, 0))[0] == 2 == O.geom_get_2dt(nan, ol, ocam, O.Geom2dtParams(2, 0.02, 0, 0, 0))[0]
    assert G.get_2dt(good, gl, gcam, G.T2dParams(2, 0.0, 0, 0, 0))[0] == 1 == O.geom_get_2dt(good, ol, ocam, O.Geom2dtParams(2, 0.0, 0, 0, 0))[0]
    with pytest.raises(ValueError):
        G.get_2dt(np.ones((3, 2)), gl, gcam, G.T2dParams(2, 0.02, 0, 0, 0))


# ---- undistortion -----------------------------------------------------------------------------------------------------

def test_undistort_points_exact_and_inverts_the_brown_model():
    gcam, ocam = cam_pair()
    rng = np.random.default_rng(5)
    uv = np.stack([rng.uniform(0, 480, 500), rng.uniform(0, 480, 500)], axis=1)
    ulx = 136.0
    got = G.undistort_points(gcam, ulx, uv)
    want = O.geom_undistort(ocam, ulx, uv)
    assert np.array_equal(got, want)
    # five iterations of cv::undistortPoints are close to, not at, the exact inverse
    cam_local = (CAM[0], CAM[1], CAM[2] - ulx) + CAM[3:]
    exact = S.undistort_exact(cam_local, uv)
    assert np.abs(got - exact).max() < 2e-2 and np.abs(S.distort(cam_local, got) - uv).max() < 1.0
    # no distortion: plain pinhole normalisation, exactly
    flat = G.Camera(CAM[0], CAM[1], CAM[2], CAM[3], 0, 0, 0, 0, 0)
    assert np.array_equal(G.undistort_points(flat, 0.0, uv), np.stack([(uv[:, 0] - CAM[2]) * (1.0 / CAM[0]), (uv[:, 1] - CAM[3]) * (1.0 / CAM[1])], axis=1))


# ---- homography decomposition --------------------------------------------------------------------------------------------

def test_decomposition_recovers_planted_motion_and_equals_oracle():
    rng = np.random.default_rng(9)
    for trial in range(100):
        R = S.rot_axis_angle(rng.normal(size=3), rng.uniform(0.001, 0.2))
        n = np.array([rng.normal(0, 0.1), rng.normal(0, 0.1), 1.0])
        n /= np.linalg.norm(n)
        t = rng.normal(0, 0.05, 3)
        H = S.plane_homography(R, t, n, 1.0) * rng.uniform(0.3, 3.0)   # any scale: removeScale() undoes it
        Rs, ts, ns = G.decompose_homography(H)
        wR, wt, wn = O.geom_decompose(H)
        assert Rs.shape[0] == 4 and np.array_equal(Rs, wR) and np.array_equal(ts, wt) and np.array_equal(ns, wn)
        err = [np.abs(Rs[k] - R).max() + np.abs(ts[k] - t).max() + np.abs(ns[k] - n).max() for k in range(4)]
        assert min(err) < 1e-8, (trial, err)
        for k in range(4):   # every solution reproduces H up to scale and is a proper rotation
            Hk = Rs[k] + np.outer(ts[k], ns[k])
            assert np.allclose(Hk / Hk[2, 2], H / H[2, 2], atol=1e-9)
            assert abs(np.linalg.det(Rs[k]) - 1) < 1e-9
    # a pure rotation gives ONE solution: R = H / s2, t = n = 0 (homography_decomp.cpp: |H'H - I|_inf < 1e-3)
    R = S.rot_axis_angle([0.2, -0.1, 1.0], 0.05)
    Rs, ts, ns = G.decompose_homography(2.5 * R)
    assert Rs.shape[0] == 1 and np.allclose(Rs[0], R, atol=1e-12) and not ts.any() and not ns.any()
    assert np.array_equal(Rs, O.geom_decompose(2.5 * R)[0])
    assert G.decompose_homography(np.zeros((3, 3)))[0].shape[0] == 0


# ---- RANSAC homography -----------------------------------------------------------------------------------------------

def _planted(rng, n_pts, outlier_frac, noise=0.0):
    R = S.rot_axis_angle(rng.normal(size=3), rng.uniform(0.002, 0.05))
    H = S.plane_homography(R, rng.normal(0, 0.03, 3), np.array([0.02, -0.01, 1.0]), 1.0)
    a = rng.uniform(-0.7, 0.7, (n_pts, 2))
    h = np.concatenate([a, np.ones((n_pts, 1))], axis=1) @ H.T
    b = h[:, :2] / h[:, 2:3] + rng.normal(0, noise, (n_pts, 2)) if noise else h[:, :2] / h[:, 2:3]
    out = rng.random(n_pts) < outlier_frac
    out[:4] = False
    sign = rng.choice([-1.0, 1.0], (n_pts, 2))
    b[out] += sign[out] * rng.uniform(0.03, 0.3, (int(out.sum()), 2))   # >= 3x the 0.01 threshold in each axis
    return a, b, ~out, H / H[2, 2]


@pytest.mark.parametrize("n_pts,frac", [(16, 0.0), (16, 0.25), (64, 0.3), (256, 0.45), (9, 0.2), (5, 0.0)])
def test_ransac_consensus_set_and_refined_homography(n_pts, frac):
    rng = np.random.default_rng(100 + n_pts)
    for trial in range(20):
        a, b, inl, Htrue = _planted(rng, n_pts, frac)
        H, mask = G.find_homography(a, b)
        wH, wmask = O.geom_find_homography(a, b)
        assert H is not None and wH is not None
        assert np.array_equal(mask, wmask)
        assert np.array_equal(mask.astype(bool), inl), (trial, mask, inl)
        assert np.allclose(H, wH, rtol=0, atol=1e-9)
        assert np.allclose(H, Htrue, rtol=0, atol=1e-7)


def test_ransac_with_measurement_noise_and_small_sets():
    rng = np.random.default_rng(77)
    a, b, inl, Htrue = _planted(rng, 64, 0.2, noise=5e-4)
    H, mask = G.find_homography(a, b)
    wH, wmask = O.geom_find_homography(a, b)
    assert np.array_equal(mask, wmask) and np.array_equal(mask.astype(bool), inl)
    assert np.allclose(H, wH, atol=1e-9) and np.allclose(H, Htrue, atol=5e-3)
    # exactly four points: solved directly, all inliers (cv::findHomography does the same)
    H4, m4 = G.find_homography(a[:4], b[:4])
    assert m4.tolist() == [1, 1, 1, 1] and np.allclose(H4, O.geom_find_homography(a[:4], b[:4])[0], atol=1e-12)
    h = np.concatenate([a[:4], np.ones((4, 1))], axis=1) @ H4.T
    assert np.allclose(h[:, :2] / h[:, 2:3], b[:4], atol=1e-10)
    # fewer than four: no model
    assert G.find_homography(a[:3], b[:3])[0] is None and O.geom_find_homography(a[:3], b[:3])[0] is None
    # all points collinear: no non-degenerate minimal set exists
    line = np.stack([np.linspace(-0.5, 0.5, 12), np.linspace(-0.5, 0.5, 12) * 0.3], axis=1)
    assert G.find_homography(line, line + 0.01)[0] is None and O.geom_find_homography(line, line + 0.01)[0] is None


# ---- getRT end to end ---------------------------------------------------------------------------------------------------

def _rt_scene(rng, gx, gy, origin, stride, patch, outliers=0, nans=0, axis=None, angle=None, dt=0.02, height=2.5, ulx=136.0,
              cam=CAM):
    axis = rng.normal(size=3) if axis is None else np.asarray(axis, float)
    angle = rng.uniform(0.002, 0.02) if angle is None else angle
    R = S.rot_axis_angle(axis, angle)
    t_metric = rng.normal(0, 0.02, 3)
    n = np.array([0.0, 0.0, 1.0])
    H = S.plane_homography(R, t_metric, n, height)
    centres = S.patch_centres(gx, gy, origin, stride, patch)
    shifts = S.shifts_for_motion(cam, ulx, centres, H)
    idx = rng.permutation(gx * gy)
    bad = idx[:outliers]
    shifts[bad] += rng.choice([-1.0, 1.0], (outliers, 2)) * rng.uniform(12, 40, (outliers, 2))  # > 0.01 * fx = 3.4 px
    shifts[idx[outliers:outliers + nans]] = np.nan
    rate = (axis / np.linalg.norm(axis)) * angle / dt
    return shifts, R, t_metric, rate, set(bad.tolist()), set(idx[outliers:outliers + nans].tolist())


# the reference's 480^2 crop of a 752x480 camera (ulCorner.x = 136), and the BASELINE c2 / c4 layouts on whole frames with
# cameras to match (the node only ever uses the first form)
@pytest.mark.parametrize("geometry", [
    (4, 4, (0, 0), (120, 120), 120, 136.0, CAM),
    (8, 8, (1, 1), (98, 59), 64, 0.0, (520.0, 518.0, 376.0, 240.0, -0.12, 0.03, 0.0004, -0.0003, -0.002)),
    (16, 16, (0, 0), (119, 63), 128, 0.0, (1300.0, 1295.0, 960.0, 540.0, -0.12, 0.03, 0.0004, -0.0003, -0.002))])
def test_get_rt_matches_oracle_and_recovers_the_motion(geometry):
    gx, gy, origin, stride, patch, ulx, cam = geometry
    rng = np.random.default_rng(gx * 31 + 1)
    gcam, ocam = G.Camera(*cam), O.GeomCamera(*cam)
    gl, ol = layouts(gx, gy, origin, stride, patch)
    n_ok = 0
    for trial in range(30):
        total = gx * gy
        shifts, R, t_metric, rate, bad, nan_idx = _rt_scene(rng, gx, gy, origin, stride, patch,
                                                            outliers=(trial % 4) * total // 16, nans=(trial % 3) * total // 16,
                                                            ulx=ulx, cam=cam)
        dt, height = 0.02, 2.5
        # the IMU quaternion the node forms: setRPY of the gyro rates (optic_flow.cpp:1314). The reference's rotation
        # estimate is the quaternion of the TRANSPOSED matrix (cvMat33ToTf2Mat33, :76-85), i.e. of the inverse motion
        ang_q = O.geom_quat_from_rpy(*(-rate))
        gp, op = rt_params(height, dt, ulx, ang_q)
        st, rot, tran, mask, H = G.get_rt(shifts, gl, gcam, gp, 8)
        wst, wrot, wtran, wmask, wH = O.geom_get_rt(shifts, ol, ocam, op, 8)
        assert st == wst and np.array_equal(mask, wmask), (trial, st, wst)
        assert np.allclose(rot, wrot, rtol=0, atol=1e-9) and np.allclose(tran, wtran, rtol=0, atol=1e-9)
        assert np.allclose(H, wH, rtol=0, atol=1e-9)
        if st != 0:
            continue
        n_ok += 1
        valid = np.array([k not in nan_idx for k in range(total)])
        assert np.array_equal(mask.astype(bool), valid & np.array([k not in bad for k in range(total)]))
        # physical meaning: |angle of o_rot| = rotation angle / dt, translation = rotated t * height / d / dt with d = height
        ang = 2 * np.arccos(np.clip(rot[3], -1, 1))
        assert abs(ang - np.linalg.norm(rate)) < 0.1 * max(1.0, np.linalg.norm(rate))   # sanity only: 5-iteration undistortion bias
        assert abs(np.linalg.norm(tran) - np.linalg.norm(t_metric) / dt) < 0.3 * np.linalg.norm(t_metric) / dt + 0.05
    assert n_ok >= 20


def test_get_rt_early_returns_and_statuses():
    rng = np.random.default_rng(4)
    gcam, ocam = cam_pair()
    gl, ol = layouts(4, 4, (0, 0), (120, 120), 120)
    shifts, R, t, rate, _, _ = _rt_scene(rng, 4, 4, (0, 0), (120, 120), 120)
    ang_q = O.geom_quat_from_rpy(*(-rate))

    def both(sh, thr=8, dt=0.02, q=ang_q):
        gp, op = rt_params(2.5, dt, 136.0, q)
        a = G.get_rt(sh, gl, gcam, gp, thr)
        b = O.geom_get_rt(sh, ol, ocam, op, thr)
        assert a[0] == b[0] and np.allclose(a[1], b[1], atol=1e-9) and np.allclose(a[2], b[2], atol=1e-9)
        return a

    assert both(shifts)[0] == 0
    assert both(shifts, dt=0.0)[0] == 1                      # !isfinite(1/dt)            :516-519
    few = shifts.copy(); few[:9] = np.nan
    assert both(few)[0] == 2                                 # 7 valid < shifted_pts_thr  :544-547
    assert both(shifts, thr=-1)[0] == 2                      # uint(-1) can never be reached
    junk = rng.uniform(-50, 50, shifts.shape)
    assert both(junk)[0] == 3                                # no consensus of 8          :575-578
    away = O.geom_quat_from_rpy(0.0, 0.0, 3.0)               # IMU says something else entirely
    assert both(shifts, q=away)[0] == 4                      # > pi/4                     :682-685
    st, rot, tran, _, _ = both(np.zeros_like(shifts), q=(0, 0, 0, 1))
    assert st == 0 and np.allclose(rot, [0, 0, 0, 1], atol=1e-9) and np.allclose(tran, 0, atol=1e-12)  # H = I: the single-solution branch :756
    fail = both(junk)
    assert np.array_equal(fail[1], [0, 0, 0, 1]) and not fail[2].any()   # outputs stay identity / zero on failure
    # cam -> base transform with a translation: `tempTfC2B * axis` applies all of it (:643)
    gp, op = rt_params(2.5, 0.02, 136.0, ang_q, c2b_q=tuple(O.geom_quat_from_rpy(0.1, -0.2, 1.5)), c2b_t=(0.05, 0.0, -0.1))
    a, b = G.get_rt(shifts, gl, gcam, gp, 8), O.geom_get_rt(shifts, ol, ocam, op, 8)
    assert a[0] == b[0] and np.allclose(a[1], b[1], atol=1e-9) and np.allclose(a[2], b[2], atol=1e-9)


def test_argument_checks():
    from mrs_optic_flow_amd import MofError
    gcam, _ = cam_pair()
    with pytest.raises(MofError):
        G.get_rt(np.zeros((2048, 2)), G.Layout(64, 32, 0, 0, 8, 8, 8), gcam, G.RtParams(), 8)   # > 1024 patches
    with pytest.raises(MofError):
        G.get_2dt(np.zeros((1, 2)), G.Layout(1, 1, 0, 0, 120, 120, 120), G.Camera(0, 1, 0, 0, 0, 0, 0, 0, 0), G.T2dParams(1, 1, 0, 0, 0))
    L = G.reference_layout(480, 120)
    assert (L.grid_x, L.grid_y, L.stride_x, L.patch_size) == (4, 4, 120, 120)


def test_unrolled_elimination_equals_the_indexed_form_bit_for_bit():
    """The device's RANSAC hypotheses solve their 8 x 9 minimal-set system with solve_linear_unrolled (every index a
    compile-time constant: the matrix stays in registers), the host form and the oracle-facing tests with solve_linear.
    tests/cpp/test_geom_core.cpp runs both on 4500 random systems (pivoting, singular and exact-zero cases) on the
    host and demands identical bits."""
    import os
    import subprocess

    here = os.path.dirname(os.path.abspath(__file__))
    binary = os.path.join(here, "cpp", "test_geom_core")
    if not os.path.exists(binary):
        subprocess.check_call(["make", "-C", os.path.join(here, "cpp"), "-s", "test_geom_core"])
    r = subprocess.run([binary], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.startswith("geom core ok"), r.stdout + r.stderr

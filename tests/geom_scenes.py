"""Synthetic planar scenes for the geometry-tail tests (test helper): a downward-looking camera with Brown
distortion above a ground plane, moved by a small rigid motion between two frames; the per-patch shifts the FFT path
would report are computed from the exact plane-induced homography."""
import numpy as np


def rot_axis_angle(axis, angle):
    a = np.asarray(axis, float)
    a = a / np.linalg.norm(a)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * K @ K


def distort(cam, xy):
    """Brown model forward: normalised (x, y) -> pixel (u, v). cam = (fx, fy, cx, cy, k1, k2, p1, p2, k3)."""
    fx, fy, cx, cy, k1, k2, p1, p2, k3 = cam
    x, y = xy[..., 0], xy[..., 1]
    r2 = x * x + y * y
    rad = 1 + ((k3 * r2 + k2) * r2 + k1) * r2
    xd = x * rad + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
    yd = y * rad + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
    return np.stack([xd * fx + cx, yd * fy + cy], axis=-1)


def undistort_exact(cam, uv, iters=60):
    """Inverse of `distort` by fixed-point iteration run to convergence (NOT OpenCV's 5 iterations)."""
    fx, fy, cx, cy, k1, k2, p1, p2, k3 = cam
    x0 = (uv[..., 0] - cx) / fx
    y0 = (uv[..., 1] - cy) / fy
    x, y = x0.copy(), y0.copy()
    for _ in range(iters):
        r2 = x * x + y * y
        ic = 1.0 / (1 + ((k3 * r2 + k2) * r2 + k1) * r2)
        dx = 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
        dy = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
        x, y = (x0 - dx) * ic, (y0 - dy) * ic
    return np.stack([x, y], axis=-1)


def plane_homography(R, t, n, d):
    """x2 ~ (R + t n' / d) x1 for points on the plane n . X = d (camera-1 frame); X2 = R X1 + t."""
    return R + np.outer(t, n) / d


def patch_centres(grid_x, grid_y, origin, stride, patch):
    i, j = np.meshgrid(np.arange(grid_x), np.arange(grid_y))
    return np.stack([origin[0] + i * stride[0] + patch // 2, origin[1] + j * stride[1] + patch // 2], axis=-1).reshape(-1, 2).astype(float)


def shifts_for_motion(cam, ul_corner_x, centres, H):
    """Pixel shifts of the patch centres under the homography H acting on normalised coordinates."""
    cam_local = (cam[0], cam[1], cam[2] - ul_corner_x) + tuple(cam[3:])
    x1 = undistort_exact(cam_local, centres)
    h = np.concatenate([x1, np.ones((x1.shape[0], 1))], axis=1) @ H.T
    x2 = h[:, :2] / h[:, 2:3]
    return distort(cam_local, x2) - centres

"""Geometry tail behind the C ABI (mof_geom_* in include/mof.h): OpticFlow::getRT and OpticFlow::get2DT
(/root/reference/src/optic_flow.cpp:515-774, :388-510) -- per-patch shifts -> camera-frame velocity.

Host forms take numpy arrays (what ``FftMethod.processImage`` returns); the batched forms take torch device tensors
(what ``process_batch_device`` returns) and stay on the GPU. Plumbing only: all arithmetic is in the library.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi
from ._capi import check

STATUS = {0: "ok", 1: "bad duration", 2: "too few points", 3: "too few inliers", 4: "angle too large",
          5: "single solution, no match", 6: "single solution, non-finite", 7: "unclassified", 8: "no homography",
          9: "no points"}


class Camera(C.Structure):
    """camMatrix_ / distCoeffs_ (optic_flow.cpp:1511-1522)."""
    _fields_ = [(n, C.c_double) for n in ("fx", "fy", "cx", "cy", "k1", "k2", "p1", "p2", "k3")]


class Layout(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("grid_x", "grid_y", "origin_x", "origin_y", "stride_x", "stride_y", "patch_size")]


class RtParams(C.Structure):
    _fields_ = [("height", C.c_double), ("dt", C.c_double), ("ul_corner_x", C.c_double),
                ("ang_rate_q", C.c_double * 4), ("c2b_q", C.c_double * 4), ("c2b_t", C.c_double * 3)]


class T2dParams(C.Structure):
    _fields_ = [("height", C.c_double), ("dt", C.c_double), ("roll_rate", C.c_double), ("pitch_rate", C.c_double),
                ("cam_yaw", C.c_double)]


RT_PARAMS_DOUBLES = C.sizeof(RtParams) // 8   # 14: one row of the device parameter array
T2D_PARAMS_DOUBLES = C.sizeof(T2dParams) // 8  # 5


def reference_layout(frame_size: int, sample_point_size: int) -> Layout:
    L = Layout()
    check(_capi.load().mof_geom_layout_reference(C.byref(L), frame_size, sample_point_size))
    return L


def _shifts(shifts, layout: Layout) -> np.ndarray:
    s = np.ascontiguousarray(shifts, dtype=np.float64)
    if s.size != 2 * layout.grid_x * layout.grid_y:
        raise ValueError(f"shifts has {s.size} values, layout needs {2 * layout.grid_x * layout.grid_y}")
    return s


def undistort_points(cam: Camera, ul_corner_x: float, pts) -> np.ndarray:
    p = np.ascontiguousarray(pts, dtype=np.float64).reshape(-1, 2)
    out = np.empty_like(p)
    check(_capi.load().mof_geom_undistort_points(C.byref(cam), float(ul_corner_x), p.ctypes.data, p.shape[0], out.ctypes.data))
    return out


def find_homography(a, b):
    """cv::findHomography(a, b, RANSAC, 0.01, mask) -> (H [3,3] or None, mask [n] uint8)."""
    a = np.ascontiguousarray(a, dtype=np.float64).reshape(-1, 2)
    b = np.ascontiguousarray(b, dtype=np.float64).reshape(-1, 2)
    if a.shape != b.shape:
        raise ValueError("a and b must hold the same number of points")
    H = np.zeros(9)
    mask = np.zeros(max(a.shape[0], 1), np.uint8)
    found = C.c_int(0)
    check(_capi.load().mof_geom_find_homography(a.ctypes.data, b.ctypes.data, a.shape[0], H.ctypes.data, mask.ctypes.data,
                                                C.byref(found)))
    return (H.reshape(3, 3) if found.value else None), mask[:a.shape[0]]


def decompose_homography(H):
    """cv::decomposeHomographyMat(H, I) -> (R [k,3,3], t [k,3], n [k,3]) with k = 1 or 4 (0 when degenerate)."""
    H = np.ascontiguousarray(H, dtype=np.float64).reshape(9)
    R, t, n = np.zeros(36), np.zeros(12), np.zeros(12)
    k = C.c_int(0)
    check(_capi.load().mof_geom_decompose_homography(H.ctypes.data, R.ctypes.data, t.ctypes.data, n.ctypes.data, C.byref(k)))
    k = k.value
    return R.reshape(4, 3, 3)[:k], t.reshape(4, 3)[:k], n.reshape(4, 3)[:k]


def get_rt(shifts, layout: Layout, cam: Camera, params: RtParams, shifted_pts_thr: int = 8):
    """OpticFlow::getRT -> (status, rot (x, y, z, w), tran (3), inlier mask [gy*gx], H [3,3])."""
    s = _shifts(shifts, layout)
    out = np.zeros(7)
    mask = np.zeros(layout.grid_x * layout.grid_y, np.uint8)
    H = np.zeros(9)
    status = C.c_int(-1)
    check(_capi.load().mof_geom_get_rt(s.ctypes.data, C.byref(layout), C.byref(cam), C.byref(params), int(shifted_pts_thr),
                                       out.ctypes.data, C.byref(status), mask.ctypes.data, H.ctypes.data))
    return status.value, out[:4].copy(), out[4:].copy(), mask, H.reshape(3, 3)


def get_2dt(shifts, layout: Layout, cam: Camera, params: T2dParams):
    """OpticFlow::get2DT -> (status, o_tran (3), o_tran_diff (3))."""
    s = _shifts(shifts, layout)
    out = np.zeros(6)
    status = C.c_int(-1)
    check(_capi.load().mof_geom_get_2dt(s.ctypes.data, C.byref(layout), C.byref(cam), C.byref(params), out.ctypes.data,
                                        C.byref(status)))
    return status.value, out[:3].copy(), out[3:].copy()


def _device_args(shifts, layout: Layout, params, row: int):
    import torch

    for name, t in (("shifts", shifts), ("params", params)):
        if not isinstance(t, torch.Tensor) or t.dtype != torch.float64 or not t.is_cuda or not t.is_contiguous():
            raise ValueError(f"{name} must be a dense float64 tensor on the GPU")
    n = shifts.shape[0] if shifts.dim() == 3 else -1
    if shifts.dim() != 3 or tuple(shifts.shape[1:]) != (layout.grid_x * layout.grid_y, 2):
        raise ValueError(f"shifts must be [n, {layout.grid_x * layout.grid_y}, 2], got {tuple(shifts.shape)}")
    if tuple(params.shape) != (n, row) or params.device != shifts.device:
        raise ValueError(f"params must be [{n}, {row}] on the same device")
    return n


def get_rt_batch_device(shifts, layout: Layout, cam: Camera, params, shifted_pts_thr: int = 8, stream=None):
    """shifts: torch float64 [n, gy*gx, 2]; params: float64 [n, 14] rows laid out as RtParams. -> float64 [n, 8] =
    rot (4), tran (3), status. Asynchronous on torch's current stream."""
    import torch

    n = _device_args(shifts, layout, params, RT_PARAMS_DOUBLES)
    out = torch.empty((n, 8), dtype=torch.float64, device=shifts.device)
    s = stream if stream is not None else torch.cuda.current_stream(shifts.device)
    check(_capi.load().mof_geom_get_rt_batch_device(shifts.data_ptr(), C.byref(layout), C.byref(cam), params.data_ptr(), n,
                                                    int(shifted_pts_thr), out.data_ptr(), C.c_void_p(s.cuda_stream)))
    return out


def get_2dt_batch_device(shifts, layout: Layout, cam: Camera, params, stream=None):
    """shifts: torch float64 [n, gy*gx, 2]; params: float64 [n, 5] rows laid out as T2dParams. -> float64 [n, 8] =
    tran (3), diff (3), status, 0."""
    import torch

    n = _device_args(shifts, layout, params, T2D_PARAMS_DOUBLES)
    out = torch.empty((n, 8), dtype=torch.float64, device=shifts.device)
    s = stream if stream is not None else torch.cuda.current_stream(shifts.device)
    check(_capi.load().mof_geom_get_2dt_batch_device(shifts.data_ptr(), C.byref(layout), C.byref(cam), params.data_ptr(), n,
                                                     out.data_ptr(), C.c_void_p(s.cuda_stream)))
    return out

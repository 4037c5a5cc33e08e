"""ctypes access to oracle/liboracle.so -- the CPU restatement used as the parity checker.

Test infrastructure only: tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg are the only callers (see oracle/oracle.h). Nothing here reads
/root/reference at run time.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "liboracle.so")


class FftLayout(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("patch", C.c_int),
                ("grid_x", C.c_int), ("grid_y", C.c_int),
                ("origin_x", C.c_int), ("origin_y", C.c_int),
                ("stride_x", C.c_int), ("stride_y", C.c_int),
                ("max_px_speed", C.c_double)]


class PcDiag(C.Structure):
    _fields_ = [("peak_x", C.c_int), ("peak_y", C.c_int), ("peak_value", C.c_double),
                ("second_value", C.c_double), ("response", C.c_double)]


class GeomCamera(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("fx", "fy", "cx", "cy", "k1", "k2", "p1", "p2", "k3")]


class GeomLayout(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("grid_x", "grid_y", "origin_x", "origin_y", "stride_x", "stride_y", "patch")]


class GeomRtParams(C.Structure):
    _fields_ = [("height", C.c_double), ("dt", C.c_double), ("ul_corner_x", C.c_double),
                ("ang_rate_q", C.c_double * 4), ("c2b_q", C.c_double * 4), ("c2b_t", C.c_double * 3)]


class Geom2dtParams(C.Structure):
    _fields_ = [("height", C.c_double), ("dt", C.c_double), ("roll_rate", C.c_double), ("pitch_rate", C.c_double),
                ("cam_yaw", C.c_double)]


class BmConfig(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("block", C.c_int), ("step", C.c_int),
                ("radius", C.c_int), ("grid_x", C.c_int), ("grid_y", C.c_int),
                ("low_contrast_rule", C.c_int)]


_lib = None


def build(force: bool = False) -> str:
    if force or not os.path.exists(ORACLE_SO):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"] + (["-B"] if force else []))
    return ORACLE_SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(ORACLE_SO)
        L.oracle_optimal_dft_size.restype = C.c_int
        L.oracle_optimal_dft_size.argtypes = [C.c_int]
        L.oracle_phase_correlate_f32.restype = C.c_int
        L.oracle_phase_correlate_f32.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int,
                                                  C.c_void_p, C.POINTER(PcDiag), C.c_void_p]
        L.oracle_phase_correlate_f64.restype = C.c_int
        L.oracle_phase_correlate_f64.argtypes = L.oracle_phase_correlate_f32.argtypes
        L.oracle_fft_process_u8.restype = C.c_int
        L.oracle_fft_process_u8.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(FftLayout), C.c_int,
                                            C.c_void_p, C.POINTER(C.c_int), C.c_void_p]
        L.oracle_phase_correlate_ocl_f32.restype = C.c_int
        L.oracle_phase_correlate_ocl_f32.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_int,
                                                      C.c_int, C.c_int, C.c_void_p, C.POINTER(PcDiag), C.c_void_p]
        L.oracle_phase_correlate_ocl_f64.restype = C.c_int
        L.oracle_phase_correlate_ocl_f64.argtypes = L.oracle_phase_correlate_ocl_f32.argtypes
        L.oracle_fft_process_ocl_u8.restype = C.c_int
        L.oracle_fft_process_ocl_u8.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(FftLayout), C.c_int, C.c_int,
                                                C.c_void_p, C.POINTER(C.c_int), C.c_void_p]
        L.oracle_resize_quarter_u8.restype = C.c_int
        L.oracle_resize_quarter_u8.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p]
        L.oracle_fft_process_long_range_u8.restype = C.c_int
        L.oracle_fft_process_long_range_u8.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(FftLayout), C.c_int,
                                                       C.c_void_p, C.POINTER(C.c_int)]
        L.oracle_rgb2gray_u8.restype = C.c_int
        L.oracle_rgb2gray_u8.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p]
        L.oracle_logpolar_u8.restype = C.c_int
        L.oracle_logpolar_u8.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_double, C.c_int, C.c_void_p]
        L.oracle_scale_rotation_step.restype = C.c_int
        L.oracle_scale_rotation_step.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_double, C.c_int, C.c_void_p,
                                                 C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.oracle_bm_config_block_method.argtypes = [C.POINTER(BmConfig), C.c_int, C.c_int, C.c_int]
        L.oracle_bm_config_fast_spaced.argtypes = [C.POINTER(BmConfig), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.oracle_bm_process_u8.restype = C.c_int
        L.oracle_bm_process_u8.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(BmConfig),
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_resize_2x_u8.restype = C.c_int
        L.oracle_resize_2x_u8.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p]
        L.oracle_bm_refine_u8.restype = C.c_int
        L.oracle_bm_refine_u8.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.c_int, C.c_void_p, C.c_void_p]
        L.oracle_bm_histogram_top.restype = C.c_int
        L.oracle_bm_histogram_top.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.oracle_logpolar_variant_u8.restype = C.c_int
        L.oracle_logpolar_variant_u8.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_double, C.c_int, C.c_int, C.c_void_p]
        L.oracle_logpolar_maps.restype = C.c_int
        L.oracle_logpolar_maps.argtypes = [C.c_int, C.c_double, C.c_int, C.c_void_p, C.c_void_p]
        L.oracle_scale_rotation_step_variant.restype = C.c_int
        L.oracle_scale_rotation_step_variant.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_double, C.c_int, C.c_void_p,
                                                         C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.oracle_undistort_point.restype = None
        L.oracle_undistort_point.argtypes = [C.POINTER(GeomCamera), C.c_double, C.c_double, C.c_double,
                                             C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.oracle_find_homography.restype = C.c_int
        L.oracle_find_homography.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.oracle_decompose_homography.restype = C.c_int
        L.oracle_decompose_homography.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_quat_from_rpy.restype = None
        L.oracle_quat_from_rpy.argtypes = [C.c_double, C.c_double, C.c_double, C.c_void_p]
        L.oracle_get_rt.restype = C.c_int
        L.oracle_get_rt.argtypes = [C.c_void_p, C.POINTER(GeomLayout), C.POINTER(GeomCamera), C.POINTER(GeomRtParams), C.c_int,
                                    C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_get_2dt.restype = C.c_int
        L.oracle_get_2dt.argtypes = [C.c_void_p, C.POINTER(GeomLayout), C.POINTER(GeomCamera), C.POINTER(Geom2dtParams),
                                     C.c_void_p]
        L.oracle_version.restype = C.c_char_p
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def optimal_dft_size(n: int) -> int:
    """cv::getOptimalDFTSize as the oracle restates it."""
    return int(lib().oracle_optimal_dft_size(int(n)))


def phase_correlate(a: np.ndarray, b: np.ndarray, precision: int = 32, want_surface: bool = False):
    """cv::phaseCorrelate(a, b) restated. Returns ((x, y), diag dict[, surface])."""
    n = a.shape[0]
    assert a.shape == (n, n) and b.shape == (n, n)
    dt = np.float32 if precision == 32 else np.float64
    a = np.ascontiguousarray(a, dtype=dt)
    b = np.ascontiguousarray(b, dtype=dt)
    out = np.zeros(2, np.float64)
    diag = PcDiag()
    m = optimal_dft_size(n)  # the surface lives on the zero-padded image
    surf = np.zeros((m, m), dt) if want_surface else None
    fn = lib().oracle_phase_correlate_f32 if precision == 32 else lib().oracle_phase_correlate_f64
    rc = fn(_ptr(a), n, _ptr(b), n, n, _ptr(out), C.byref(diag), _ptr(surf) if want_surface else None)
    if rc:
        raise ValueError(f"oracle_phase_correlate rc={rc}")
    d = dict(peak=(diag.peak_x, diag.peak_y), peak_value=diag.peak_value, second_value=diag.second_value,
             response=diag.response)
    if want_surface:
        return (out[0], out[1]), d, surf
    return (out[0], out[1]), d


def f32_floor_bins(a: np.ndarray, b: np.ndarray) -> int:
    """Number of spectral bins of EITHER patch (zero-padded to cv::phaseCorrelate's transform size) that are zero in exact
    arithmetic, i.e. lie below the rounding floor of any f32 transform (32 eps32 ||x||_2). What an f32 transform leaves in
    such a bin -- an exact zero or 1e-7-relative noise, which the cross-power normalisation then blows up to unit magnitude --
    depends on its radix order; the reference's value there is an accident of OpenCV's factorisation, and each such bin pair
    moves the centroid by about 1 / (M^2 peak) px. A few of them occur by coincidence in integer data (three residue-class
    sums equal at a bin made of cube roots of unity: found by tools/fft_sr_fuzz.py seed 20261004); constant patches have
    nothing else. Test-side diagnostic (numpy), not part of the C oracle."""
    m = optimal_dft_size(a.shape[0])
    n = 0
    for x in (a, b):
        xf = x.astype(np.float64)
        spec = np.abs(np.fft.fft2(xf, s=(m, m)))
        n += int((spec < 32.0 * float(np.finfo(np.float32).eps) * np.sqrt((xf * xf).sum())).sum())
    return n


def fft_layout(width, height, patch, grid_x, grid_y, origin=(0, 0), stride=None, max_px_speed=80.0) -> FftLayout:
    stride = stride or (patch, patch)
    return FftLayout(width, height, patch, grid_x, grid_y, origin[0], origin[1], stride[0], stride[1],
                     float(max_px_speed))


def fft_process(cur: np.ndarray, prev: np.ndarray, layout: FftLayout, precision: int = 32, want_diag: bool = False):
    """FftMethod::processImage (useOCL=false) on one uint8 frame pair -> [gy*gx, 2] float64."""
    assert cur.dtype == np.uint8 and prev.dtype == np.uint8 and cur.shape == prev.shape
    cur = np.ascontiguousarray(cur)
    prev = np.ascontiguousarray(prev)
    g = layout.grid_x * layout.grid_y
    out = np.zeros((g, 2), np.float64)
    ninv = C.c_int(0)
    diags = (PcDiag * g)() if want_diag else None
    rc = lib().oracle_fft_process_u8(_ptr(cur), _ptr(prev), cur.shape[1], C.byref(layout), precision, _ptr(out),
                                     C.byref(ninv), diags)
    if rc:
        raise ValueError(f"oracle_fft_process_u8 rc={rc}")
    if want_diag:
        return out, ninv.value, diags
    return out, ninv.value


def phase_correlate_ocl(a: np.ndarray, b: np.ndarray, origin=(0, 0), search_radius: int = 55, precision: int = 32,
                        want_surface: bool = False):
    """The useOCL=true peak model on one patch pair. Returns ((sx, sy), diag dict[, surface]) -- the SHIFT itself."""
    n = a.shape[0]
    assert a.shape == (n, n) and b.shape == (n, n)
    dt = np.float32 if precision == 32 else np.float64
    a = np.ascontiguousarray(a, dtype=dt)
    b = np.ascontiguousarray(b, dtype=dt)
    out = np.zeros(2, np.float64)
    diag = PcDiag()
    surf = np.zeros((n, n), dt) if want_surface else None
    fn = lib().oracle_phase_correlate_ocl_f32 if precision == 32 else lib().oracle_phase_correlate_ocl_f64
    rc = fn(_ptr(a), n, _ptr(b), n, n, int(origin[0]), int(origin[1]), int(search_radius), _ptr(out), C.byref(diag),
            _ptr(surf) if want_surface else None)
    if rc:
        raise ValueError(f"oracle_phase_correlate_ocl rc={rc}")
    d = dict(peak=(diag.peak_x, diag.peak_y), peak_value=diag.peak_value, second_value=diag.second_value,
             response=diag.response)
    if want_surface:
        return (out[0], out[1]), d, surf
    return (out[0], out[1]), d


def fft_process_ocl(cur: np.ndarray, prev: np.ndarray, layout: FftLayout, search_radius: int = 55, precision: int = 32,
                    want_diag: bool = False):
    """FftMethod::processImage under the useOCL=true peak model -> [gy*gx, 2] float64."""
    assert cur.dtype == np.uint8 and prev.dtype == np.uint8 and cur.shape == prev.shape
    cur = np.ascontiguousarray(cur)
    prev = np.ascontiguousarray(prev)
    g = layout.grid_x * layout.grid_y
    out = np.zeros((g, 2), np.float64)
    ninv = C.c_int(0)
    diags = (PcDiag * g)() if want_diag else None
    rc = lib().oracle_fft_process_ocl_u8(_ptr(cur), _ptr(prev), cur.shape[1], C.byref(layout), int(search_radius),
                                         precision, _ptr(out), C.byref(ninv), diags)
    if rc:
        raise ValueError(f"oracle_fft_process_ocl_u8 rc={rc}")
    if want_diag:
        return out, ninv.value, diags
    return out, ninv.value


def resize_quarter(img: np.ndarray) -> np.ndarray:
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    out = np.zeros((h // 4, w // 4), np.uint8)
    rc = lib().oracle_resize_quarter_u8(_ptr(img), w, w, h, _ptr(out))
    if rc:
        raise ValueError(f"oracle_resize_quarter_u8 rc={rc}")
    return out


def fft_process_long_range(cur: np.ndarray, prev: np.ndarray, layout: FftLayout, precision: int = 32):
    """FftMethod::processImageLongRange on one uint8 frame pair -> [(gy/4)*(gx/4), 2] float64."""
    cur = np.ascontiguousarray(cur)
    prev = np.ascontiguousarray(prev)
    g = (layout.grid_x // 4) * (layout.grid_y // 4)
    out = np.zeros((g, 2), np.float64)
    ninv = C.c_int(0)
    rc = lib().oracle_fft_process_long_range_u8(_ptr(cur), _ptr(prev), cur.shape[1], C.byref(layout), precision,
                                                _ptr(out), C.byref(ninv))
    if rc:
        raise ValueError(f"oracle_fft_process_long_range_u8 rc={rc}")
    return out, ninv.value


def rgb2gray(img: np.ndarray) -> np.ndarray:
    """cv::cvtColor(img, CV_RGB2GRAY) on an [H, W, 3] uint8 array (whatever its channel order really is)."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w, c = img.shape
    assert c == 3
    out = np.zeros((h, w), np.uint8)
    rc = lib().oracle_rgb2gray_u8(_ptr(img), 3 * w, w, h, _ptr(out))
    if rc:
        raise ValueError(f"oracle_rgb2gray_u8 rc={rc}")
    return out


def logpolar(src: np.ndarray, M: float, interp: int, dst: np.ndarray | None = None, variant: int = 0) -> np.ndarray:
    """cv::logPolar(src, dst, (res/2, res/2), M, interp) on a square uint8 image; dst is updated in place.
    variant 0 = OpenCV 4.x cv::logPolar (warpPolar form), 1 = OpenCV 3.2 cvLogPolar."""
    src = np.ascontiguousarray(src, dtype=np.uint8)
    res = src.shape[0]
    assert src.shape == (res, res)
    if dst is None:
        dst = np.zeros((res, res), np.uint8)
    assert dst.flags.c_contiguous and dst.dtype == np.uint8 and dst.shape == (res, res)
    rc = lib().oracle_logpolar_variant_u8(_ptr(src), res, res, float(M), int(interp), int(variant), _ptr(dst))
    if rc:
        raise ValueError(f"oracle_logpolar_variant_u8 rc={rc}")
    return dst


def logpolar_maps(res: int, M: float, variant: int = 0):
    """The float maps cv::logPolar hands to cv::remap: (mapx, mapy), each [res, res] float32, rows = phi."""
    mx, my = np.zeros((res, res), np.float32), np.zeros((res, res), np.float32)
    rc = lib().oracle_logpolar_maps(res, float(M), int(variant), _ptr(mx), _ptr(my))
    if rc:
        raise ValueError(f"oracle_logpolar_maps rc={rc}")
    return mx, my


class ScaleRotationEstimator:
    """scaleRotationEstimator restated (state: tempIm, prevIm_F32, first)."""

    def __init__(self, res: int, M: float, precision: int = 32, variant: int = 0):
        self.res, self.M, self.precision, self.variant = res, float(M), precision, int(variant)
        self.temp_im = np.zeros((res, res), np.uint8)
        self.prev_lp = np.zeros((res, res), np.float32)
        self.first = True
        self.pt = (0.0, 0.0)

    def processImage(self, frame: np.ndarray):
        frame = np.ascontiguousarray(frame, dtype=np.uint8)
        assert frame.shape == (self.res, self.res)
        out = np.zeros(2)
        pt = np.zeros(2)
        rc = lib().oracle_scale_rotation_step_variant(_ptr(frame), self.res, self.res, self.M, int(self.first),
                                                      _ptr(self.temp_im), _ptr(self.prev_lp), self.precision, self.variant,
                                                      _ptr(out), _ptr(pt))
        if rc:
            raise ValueError(f"oracle_scale_rotation_step rc={rc}")
        self.first = False
        self.pt = (float(pt[0]), float(pt[1]))
        return float(out[0]), float(out[1])


def bm_config_block_method(frame_size, block, radius) -> BmConfig:
    c = BmConfig()
    lib().oracle_bm_config_block_method(C.byref(c), frame_size, block, radius)
    return c


def bm_config_fast_spaced(width, height, block, step, radius) -> BmConfig:
    c = BmConfig()
    lib().oracle_bm_config_fast_spaced(C.byref(c), width, height, block, step, radius)
    return c


def bm_process(cur: np.ndarray, prev: np.ndarray, cfg: BmConfig, want_sad: bool = False):
    """Block scan on one uint8 frame pair -> dx[gy,gx], dy[gy,gx] int8, mode (x,y)[, sad_all]."""
    assert cur.dtype == np.uint8 and prev.dtype == np.uint8 and cur.shape == prev.shape
    cur = np.ascontiguousarray(cur)
    prev = np.ascontiguousarray(prev)
    g = cfg.grid_x * cfg.grid_y
    d = 2 * cfg.radius + 1
    dx = np.zeros(g, np.int8)
    dy = np.zeros(g, np.int8)
    mode = np.zeros(2, np.int8)
    sad = np.zeros((g, d, d), np.int32) if want_sad else None
    rc = lib().oracle_bm_process_u8(_ptr(cur), _ptr(prev), cur.shape[1], C.byref(cfg), _ptr(dx), _ptr(dy), _ptr(mode),
                                    None, _ptr(sad) if want_sad else None)
    if rc:
        raise ValueError(f"oracle_bm_process_u8 rc={rc}")
    res = (dx.reshape(cfg.grid_y, cfg.grid_x), dy.reshape(cfg.grid_y, cfg.grid_x), (int(mode[0]), int(mode[1])))
    return res + (sad,) if want_sad else res


def bm_histogram_top(d: np.ndarray, radius: int, depth: int = 3) -> np.ndarray:
    d = np.ascontiguousarray(d, dtype=np.int8).ravel()
    top = np.zeros(depth, np.int8)
    rc = lib().oracle_bm_histogram_top(_ptr(d), d.size, radius, depth, _ptr(top))
    if rc:
        raise ValueError(f"oracle_bm_histogram_top rc={rc}")
    return top


def resize_2x(img: np.ndarray) -> np.ndarray:
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    out = np.zeros((2 * h, 2 * w), np.uint8)
    rc = lib().oracle_resize_2x_u8(_ptr(img), w, w, h, _ptr(out))
    if rc:
        raise ValueError(f"oracle_resize_2x_u8 rc={rc}")
    return out


def bm_refine(cur: np.ndarray, prev: np.ndarray, fullpix, passes: int = 2, faithful: bool = True, want_sads: bool = False):
    """BlockMethod::Refine restated -> (x, y)[, sads[passes, 3, 3]]."""
    cur = np.ascontiguousarray(cur, dtype=np.uint8)
    prev = np.ascontiguousarray(prev, dtype=np.uint8)
    h, w = cur.shape
    out = np.zeros(2)
    sads = np.zeros((passes, 3, 3), np.int32)
    rc = lib().oracle_bm_refine_u8(_ptr(cur), _ptr(prev), w, w, h, int(fullpix[0]), int(fullpix[1]), passes, int(faithful),
                                   _ptr(out), _ptr(sads))
    if rc:
        raise ValueError(f"oracle_bm_refine_u8 rc={rc}")
    return ((float(out[0]), float(out[1])), sads) if want_sads else (float(out[0]), float(out[1]))


# ---- geometry tail (oracle/geom_ref.c) ---------------------------------------------------------------------------

def geom_undistort(cam: GeomCamera, ul_corner_x: float, pts) -> np.ndarray:
    pts = np.asarray(pts, dtype=np.float64).reshape(-1, 2)
    out = np.zeros_like(pts)
    x, y = C.c_double(), C.c_double()
    for i, (u, v) in enumerate(pts):
        lib().oracle_undistort_point(C.byref(cam), float(ul_corner_x), float(u), float(v), C.byref(x), C.byref(y))
        out[i] = (x.value, y.value)
    return out


def geom_find_homography(a, b):
    a = np.ascontiguousarray(a, dtype=np.float64).reshape(-1, 2)
    b = np.ascontiguousarray(b, dtype=np.float64).reshape(-1, 2)
    H = np.zeros(9)
    mask = np.zeros(max(a.shape[0], 1), np.uint8)
    found = lib().oracle_find_homography(_ptr(a), _ptr(b), a.shape[0], _ptr(H), _ptr(mask))
    return (H.reshape(3, 3) if found else None), mask[:a.shape[0]]


def geom_decompose(H):
    H = np.ascontiguousarray(H, dtype=np.float64).reshape(9)
    R, t, n = np.zeros(36), np.zeros(12), np.zeros(12)
    k = lib().oracle_decompose_homography(_ptr(H), _ptr(R), _ptr(t), _ptr(n))
    return R.reshape(4, 3, 3)[:k], t.reshape(4, 3)[:k], n.reshape(4, 3)[:k]


def geom_quat_from_rpy(roll, pitch, yaw) -> np.ndarray:
    q = np.zeros(4)
    lib().oracle_quat_from_rpy(float(roll), float(pitch), float(yaw), _ptr(q))
    return q


def geom_get_rt(shifts, layout: GeomLayout, cam: GeomCamera, params: GeomRtParams, thr: int = 8):
    s = np.ascontiguousarray(shifts, dtype=np.float64)
    out, H = np.zeros(7), np.zeros(9)
    mask = np.zeros(layout.grid_x * layout.grid_y, np.uint8)
    status = lib().oracle_get_rt(_ptr(s), C.byref(layout), C.byref(cam), C.byref(params), int(thr), _ptr(out), _ptr(mask), _ptr(H))
    return status, out[:4].copy(), out[4:].copy(), mask, H.reshape(3, 3)


def geom_get_2dt(shifts, layout: GeomLayout, cam: GeomCamera, params: Geom2dtParams):
    s = np.ascontiguousarray(shifts, dtype=np.float64)
    out = np.zeros(6)
    status = lib().oracle_get_2dt(_ptr(s), C.byref(layout), C.byref(cam), C.byref(params), _ptr(out))
    return status, out[:3].copy(), out[3:].copy()


# ---- the TUNED CPU path (oracle/pc_fast.c): bench.py's cpu_baseline.tuned leg, NOT the parity oracle ----------------
_fast = None


def fast_lib(path: str | None = None):
    """liboracle's sibling libpcfast.so (or a -march=native rebuild of it at `path`)."""
    global _fast
    if _fast is None or path is not None:
        p = path or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "libpcfast.so")
        lib_ = C.CDLL(p)
        lib_.pcfast_fft_process_u8.restype = C.c_int
        lib_.pcfast_fft_process_u8.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(FftLayout), C.c_void_p]
        _fast = lib_
    return _fast


def fft_process_fast(cur: np.ndarray, prev: np.ndarray, layout: FftLayout) -> np.ndarray:
    """Same estimator as fft_process(.., 32), power-of-two patches only (the tuned implementation)."""
    assert cur.dtype == np.uint8 and prev.dtype == np.uint8 and cur.shape == prev.shape
    cur = np.ascontiguousarray(cur)
    prev = np.ascontiguousarray(prev)
    out = np.zeros((layout.grid_x * layout.grid_y, 2), np.float64)
    rc = fast_lib().pcfast_fft_process_u8(cur.ctypes.data, prev.ctypes.data, cur.shape[1], C.byref(layout), out.ctypes.data)
    if rc:
        raise ValueError(f"pcfast_fft_process_u8 rc={rc}")
    return out

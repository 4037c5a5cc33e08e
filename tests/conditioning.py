"""How well does f32 arithmetic determine one patch pair's sub-pixel shift?  A criterion of the INPUTS alone.

Test infrastructure (numpy / torch on the CPU); nothing here looks at the kernel's answer or at oracle/*.c. It exists because
`-cv::phaseCorrelate` (/root/reference/src/FftMethod.cpp:1836) computes in CV_32F and normalises every cross-power bin to unit
magnitude (`divSpectrums`, :1086-1251): a bin whose value is rounding noise weighs as much as any other, so on some inputs the answer
depends on the transform's radix order -- OpenCV's, which this image does not have. Three mechanisms, each measured here:

  exact-zero bin        a bin of A or B that is zero in exact arithmetic (|X| < 1e-9 ||x||_2 in a float64 transform). An f32 transform
                        leaves either an exact 0 there (the bin then contributes nothing) or ~1e-7-relative noise (the bin is
                        normalised to unit magnitude with a random phase): which one is an accident of the factorisation. Each such
                        bin moves the 5 x 5 centroid by up to lever / |S| px (S = the window's sum).
  rounding-floor bin    bins that are small but not zero (|X| < 32 eps32 ||x||_2, smooth content): their phase is f32 noise.
  centroid cancellation the 5 x 5 window holds values of both signs and its sum S nearly cancels (sum|w| / |S| large, e.g. a constant
                        frame against texture: the surface is noise): every rounding of the surface is amplified by that ratio.

and ONE measurement that needs no model: the same pipeline through several independent f32 transforms (pocketfft complex64 rows-then-
columns, the same on the transposed patch, pocketfft's REAL transform rfft2 / irfft2, torch.fft complex64 and real) against the
float64 pipeline. `spread_px` = the largest distance of any of them from the f64 answer = how far a correct f32 implementation can
land from the truth on THIS input. tests/tolerances.py turns it into the bar.
"""
from __future__ import annotations

import numpy as np

FLT_EPS = float(np.finfo(np.float32).eps)
DBL_EPS = float(np.finfo(np.float64).eps)
ZERO_REL = 1e-9           # |X| below this x ||x||_2 in float64: zero in exact arithmetic (pocketfft f64 noise is ~1e-15 ||x||_2 sqrt(M))
FLOOR_REL = 32 * FLT_EPS  # |X| below this x ||x||_2: under the rounding floor of any f32 transform
CANCEL_FROM = 4.0         # sum|w| / |sum w| over the 5 x 5 window above which the centroid is called cancelling


def optimal_dft_size(n: int) -> int:
    m = n
    while True:
        r = m
        for p in (2, 3, 5):
            while r % p == 0:
                r //= p
        if r == 1:
            return m
        m += 1


def _slots(m):
    return (0, m // 2) if m % 2 == 0 else (0,)


def _centroid(c, m):
    """fftShift + first maximum + 5 x 5 centroid in float64 on an unscaled inverse `c` (m x m) -> (shift xy, window, S, peak)."""
    s = np.roll(np.asarray(c, np.float64), (m // 2, m // 2), axis=(0, 1))
    py, px = divmod(int(np.argmax(s)), m)
    y0, y1, x0, x1 = max(py - 2, 0), min(py + 2, m - 1), max(px - 2, 0), min(px + 2, m - 1)
    w = s[y0:y1 + 1, x0:x1 + 1]
    ys, xs = np.mgrid[y0:y1 + 1, x0:x1 + 1]
    S = float(w.sum())
    tot = S + DBL_EPS
    return np.array([(xs * w).sum() / tot - m / 2.0, (ys * w).sum() / tot - m / 2.0]), w, S, float(s[py, px]), (px, py)


def _pad(x, m, dt):
    p = np.zeros((m, m), dt)
    p[:x.shape[0], :x.shape[1]] = x
    return p


def _normalise(P, dt, m, half):
    """P |P| / (|P|^2 + eps) per bin; the real-only slots P / (P^2 + eps). `half`: P is an rfft2 half spectrum."""
    mag = np.abs(P)
    C = (P * mag / (mag * mag + dt(FLT_EPS))).astype(P.dtype)
    for r in _slots(m):
        for c in _slots(m):
            p = P[r, c].real
            C[r, c] = p / (p * p + dt(FLT_EPS))
    return C


def pipeline(a, b, how="np", dt=np.float32):
    """-cv::phaseCorrelate(a, b) through one transform library / order. Returns the shift (x, y) in float64."""
    m = optimal_dft_size(a.shape[0])
    pa, pb = _pad(a, m, dt), _pad(b, m, dt)
    if how == "np":
        A, B = np.fft.fft2(pa), np.fft.fft2(pb)
        c = np.fft.ifft2(_normalise(A * np.conj(B), dt, m, False)).real * (m * m)
    elif how == "np_t":  # columns first: the same library on the transposed patch
        A, B = np.fft.fft2(pa.T).T, np.fft.fft2(pb.T).T
        c = np.fft.ifft2(_normalise(A * np.conj(B), dt, m, False).T).T.real * (m * m)
    elif how == "np_r":  # pocketfft's real transform
        A, B = np.fft.rfft2(pa), np.fft.rfft2(pb)
        c = np.fft.irfft2(_normalise(A * np.conj(B), dt, m, True), s=(m, m)) * (m * m)
    elif how in ("torch", "torch_r"):
        import torch

        ta, tb = torch.from_numpy(pa), torch.from_numpy(pb)
        if how == "torch":
            A, B = torch.fft.fft2(ta).numpy(), torch.fft.fft2(tb).numpy()
            c = torch.fft.ifft2(torch.from_numpy(_normalise(A * np.conj(B), dt, m, False))).numpy().real * (m * m)
        else:
            A, B = torch.fft.rfft2(ta).numpy(), torch.fft.rfft2(tb).numpy()
            c = torch.fft.irfft2(torch.from_numpy(_normalise(A * np.conj(B), dt, m, True)), s=(m, m)).numpy() * (m * m)
    else:
        raise ValueError(how)
    return _centroid(c, m)[0]


F32_LIBRARIES = ("np", "np_t", "np_r", "torch", "torch_r")


def analyse(a: np.ndarray, b: np.ndarray) -> dict:
    """Everything the bars need about one patch pair (a = cur patch, b = prev patch, uint8 N x N), from the inputs alone."""
    a, b = np.asarray(a), np.asarray(b)
    m = optimal_dft_size(a.shape[0])
    zero = floor = 0
    zero_list = []
    for name, x in (("A", a), ("B", b)):
        xf = x.astype(np.float64)
        spec = np.abs(np.fft.fft2(_pad(xf, m, np.float64)))
        nrm = float(np.sqrt((xf * xf).sum()))
        z = spec < ZERO_REL * nrm if nrm > 0 else np.ones_like(spec, bool)
        zero += int(z.sum())
        floor += int(((spec < FLOOR_REL * nrm) & ~z).sum())
        if 0 < z.sum() <= 8:
            zero_list += [f"{name}[{r},{c}]" for r, c in zip(*np.nonzero(z))]
    pa, pb = _pad(a, m, np.float64), _pad(b, m, np.float64)
    A, B = np.fft.fft2(pa), np.fft.fft2(pb)
    c = np.fft.ifft2(_normalise(A * np.conj(B), np.float64, m, False)).real * (m * m)
    r64, w, S, peak, _ = _centroid(c, m)
    cancel = float(np.abs(w).sum() / max(abs(S), 1e-300))
    per_lib = {}
    for how in F32_LIBRARIES:
        try:
            per_lib[how] = float(np.abs(pipeline(a, b, how, np.float32) - r64).max())
        except ImportError:
            pass
    spread = max(per_lib.values())
    # each exact-zero bin an f32 transform fails to cancel adds a unit-magnitude term to the surface: the centroid moves by at most
    # lever / |S| (lever = the window's half diagonal, 2 sqrt 2 px)
    zero_px = zero * 2.0 * np.sqrt(2.0) / max(abs(S), 1e-300)
    # every mechanism that applies, the one with the larger lever first (a constant frame against texture has whole zero rows, which
    # every transform cancels, AND a cancelling window, which is what scatters the libraries)
    mechs = (["centroid cancellation"] if cancel > CANCEL_FROM else []) + (["exact-zero bin"] if zero > 0 else []) + \
            (["rounding-floor bin"] if floor > 0 else [])
    mech = " + ".join(mechs) if mechs else "f32 rounding (no sub-floor bin, no cancellation)"
    return {"mechanism": mech, "transform_size": m, "zero_bins": zero, "zero_bin_list": zero_list, "floor_bins": floor,
            "cancellation": cancel, "window_sum_over_m2": S / (m * m), "peak_over_m2": peak / (m * m),
            "f64_pipeline_xy": [float(r64[0]), float(r64[1])], "independent_f32_minus_f64_px": per_lib, "spread_px": spread,
            "zero_bin_px": float(zero_px)}

"""Independent numpy restatement of the hot path, used ONLY to cross-check oracle/*.c.

It shares no code with the C oracle: transforms come from numpy.fft (pocketfft,
double precision), the spectrum is processed as a full Hermitian array (no CCS
bookkeeping), block matching uses broadcasting. Written from SURVEY.md Appendix
A/B; nothing here imports or reads /root/reference.
"""
from __future__ import annotations

import numpy as np

FLT_EPS = float(np.finfo(np.float32).eps)
DBL_EPS = float(np.finfo(np.float64).eps)


def optimal_dft_size(n: int) -> int:
    """cv::getOptimalDFTSize: the smallest 2^a 3^b 5^c >= n (written from OpenCV's documentation)."""
    m = n
    while True:
        r = m
        for p in (2, 3, 5):
            while r % p == 0:
                r //= p
        if r == 1:
            return m
        m += 1


def phase_correlate(a: np.ndarray, b: np.ndarray, quirk: bool = True):
    """cv::phaseCorrelate(a, b) in float64. Returns ((x, y), surface, (px, py)). As OpenCV does, the images are zero-padded
    (bottom / right) to n = getOptimalDFTSize(size), which may be odd; surface, peak and centre live on the padded image."""
    n_in = a.shape[0]
    assert a.shape == (n_in, n_in) == b.shape
    n = optimal_dft_size(n_in)
    pa = np.zeros((n, n))
    pb = np.zeros((n, n))
    pa[:n_in, :n_in] = a
    pb[:n_in, :n_in] = b
    A = np.fft.fft2(pa)
    B = np.fft.fft2(pb)
    P = A * np.conj(B)
    mag = np.abs(P)
    Cs = P * mag / (mag * mag + FLT_EPS)
    if quirk:  # the real-only CCS slots (DC, and the Nyquist bins when n is even): P / (P^2 + eps)   
        slots = (0, n // 2) if n % 2 == 0 else (0,)
        for r in slots:
            for c in slots:
                p = P[r, c].real
                Cs[r, c] = p / (p * p + FLT_EPS)
    c = np.fft.ifft2(Cs).real * (n * n)  # cv::idft is unscaled
    s = np.roll(c, (n // 2, n // 2), axis=(0, 1))  # fftShift: index i -> (i + n // 2) % n, also for odd n
    flat = int(np.argmax(s))  # first maximum, row-major
    py, px = divmod(flat, n)
    y0, y1 = max(py - 2, 0), min(py + 2, n - 1)
    x0, x1 = max(px - 2, 0), min(px + 2, n - 1)
    cx = cy = tot = 0.0
    for y in range(y0, y1 + 1):
        for x in range(x0, x1 + 1):
            v = float(s[y, x])
            cx += x * v
            cy += y * v
            tot += v
    tot += DBL_EPS
    return (n / 2.0 - cx / tot, n / 2.0 - cy / tot), s, (px, py)


def phase_correlate_ocl(a: np.ndarray, b: np.ndarray, search_radius: int = 55):
    """The useOCL=true peak model in float64: rsqrt normalisation, 1/(ab) in the real-only slots, scaled
    inverse, +-search_radius mask, first maximum, 7x7 positive-only centroid seeded with FLT_EPSILON.
    Returns ((sx, sy), surface, (px, py)) -- the shift itself."""
    n = a.shape[0]
    assert a.shape == (n, n) == b.shape and n % 2 == 0
    A = np.fft.fft2(a.astype(np.float64))
    B = np.fft.fft2(b.astype(np.float64))
    P = A * np.conj(B)
    with np.errstate(divide="ignore", invalid="ignore"):
        Cs = P / np.sqrt(np.abs(P) ** 2 + FLT_EPS)
        h = n // 2
        for r in (0, h):
            for c in (0, h):
                Cs[r, c] = 1.0 / (A[r, c].real * B[r, c].real)
        c = np.fft.ifft2(Cs).real  # scaled by 1/N^2, as the kernel does
    idx = np.arange(n)
    masked = (idx > search_radius) & (idx < n - search_radius)
    c = np.where(masked[:, None] | masked[None, :], 0.0, c)
    s = np.fft.fftshift(c)
    best, px, py = -np.finfo(np.float32).max, 0, 0
    flat = s.ravel()
    if not np.all(np.isnan(flat)):
        m = np.nanmax(flat)
        if m > best:
            k = int(np.flatnonzero(flat == m)[0])
            py, px = divmod(k, n)
    y0, y1 = max(py - 3, 0), min(py + 3, n - 1)
    x0, x1 = max(px - 3, 0), min(px + 3, n - 1)
    cx = cy = 0.0
    tot = FLT_EPS
    for y in range(y0, y1 + 1):
        for x in range(x0, x1 + 1):
            v = float(s[y, x])
            if v > 0.0:
                cx += x * v
                cy += y * v
                tot += v
    return (cx / tot - n // 2, cy / tot - n // 2), s, (px, py)


def fft_process(cur: np.ndarray, prev: np.ndarray, patch: int, grid, origin=(0, 0), stride=None,
                max_px_speed: float = 80.0) -> np.ndarray:
    """FftMethod::processImage restated: [gy*gx, 2] float64 with NaN gating."""
    stride = stride or (patch, patch)
    gx, gy = grid
    out = np.zeros((gx * gy, 2))
    for j in range(gy):
        for i in range(gx):
            x0, y0 = origin[0] + i * stride[0], origin[1] + j * stride[1]
            (tx, ty), _, _ = phase_correlate(cur[y0:y0 + patch, x0:x0 + patch], prev[y0:y0 + patch, x0:x0 + patch])
            sx, sy = -tx, -ty
            bad = sx * sx + sy * sy > max_px_speed ** 2 or abs(sx) > patch / 2 or abs(sy) > patch / 2
            bad = bad or np.isnan(sx) or np.isnan(sy)
            out[i + j * gx] = (np.nan, np.nan) if bad else (sx, sy)
    return out


def bm_process(cur: np.ndarray, prev: np.ndarray, block: int, step: int, radius: int, grid, low_contrast: bool):
    """Block scan restated with broadcasting -> dx[gy,gx], dy[gy,gx], (modex, modey), sad[gy,gx,D,D]."""
    gx, gy = grid
    r, S, D = radius, block + step, 2 * radius + 1
    dx = np.zeros((gy, gx), np.int8)
    dy = np.zeros((gy, gx), np.int8)
    sads = np.zeros((gy, gx, D, D), np.int64)
    for by in range(gy):
        for bx in range(gx):
            cb = cur[by * S + r:by * S + r + block, bx * S + r:bx * S + r + block].astype(np.int64)
            win = prev[by * S:by * S + block + 2 * r, bx * S:bx * S + block + 2 * r].astype(np.int64)
            v = np.lib.stride_tricks.sliding_window_view(win, (block, block))  # [D, D, block, block]
            sad = np.abs(v - cb[None, None]).sum(axis=(2, 3))
            sads[by, bx] = sad
            k = int(np.argmin(sad))  # first minimum, row-major
            my, mx = divmod(k, D)
            if low_contrast and float(sad[r, r] - sad[my, mx]) <= r * r * 0.2:
                mx, my = r, r
            dx[by, bx], dy[by, bx] = mx - r, my - r
    hx = np.bincount(dx.ravel().astype(np.int64) + r, minlength=D)
    hy = np.bincount(dy.ravel().astype(np.int64) + r, minlength=D)
    return dx, dy, (int(np.argmax(hx)) - r, int(np.argmax(hy)) - r), sads

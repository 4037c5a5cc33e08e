"""Deterministic synthetic frame pairs.

The reference ships no images, bag files or fixtures, so every
workload is generated: a counter-hash texture ``T_k(y, x) = mix32(seed, k, y, x) >> 24``
(optionally 3x3 box-blurred with integer rounding), from which ``prev`` is a window
and ``cur`` is the same window displaced so that the image CONTENT moves by
``(+dx, +dy)`` pixels from prev to cur -- the sign FftMethod reports
(/root/reference/src/FftMethod.cpp:1836, ``shift = -cv::phaseCorrelate(cur, prev)``).

The same arithmetic is implemented for numpy (CPU tests, oracle inputs) and for
torch tensors on any device (bench.py generates the batch directly in HBM), and
both produce identical bytes.
"""
from __future__ import annotations

import numpy as np

SEED = 0x5EED0F10
MARGIN = 32  # canvas margin; planted shifts must satisfy |d| <= MARGIN - 1

_M32 = 0xFFFFFFFF
_K_PAIR, _K_ROW, _K_COL = 0x9E3779B1, 0x85EBCA77, 0xC2B2AE3D
_F1, _F2 = 0x85EBCA6B, 0xC2B2AE35


def planted_shift(k: int, s: int) -> tuple[int, int]:
    """Planted integer translation of pair ``k`` with amplitude ``s``."""
    return (k * 7) % (2 * s + 1) - s, (k * 13) % (2 * s + 1) - s


def _mix32_np(seed: int, k: np.ndarray, y: np.ndarray, x: np.ndarray) -> np.ndarray:
    h = (np.uint64(seed) ^ (k.astype(np.uint64) * np.uint64(_K_PAIR))) & np.uint64(_M32)
    h = (h ^ (y.astype(np.uint64) * np.uint64(_K_ROW))) & np.uint64(_M32)
    h = (h ^ (x.astype(np.uint64) * np.uint64(_K_COL))) & np.uint64(_M32)
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(_F1)) & np.uint64(_M32)
    h ^= h >> np.uint64(13)
    h = (h * np.uint64(_F2)) & np.uint64(_M32)
    h ^= h >> np.uint64(16)
    return h


def _blur_taps(blur):
    """3-tap separable integer blur: True = the 3 x 3 box of SURVEY 8(d), (sum + 4) / 9; "mild" = (1 6 1) x (1 6 1), (sum + 32) / 64.
    The box has spectral zeros (at 1/3 of the sampling rate) and leaves small alternating-sign pixel sums, which makes patches with a
    spectral bin that is EXACTLY zero -- three equal integer residue-class sums -- an every-few-hundred-patches event (tests/
    conditioning.py: no f32 arithmetic pins such a patch); "mild" has no zero (response 0.25 .. 1) and keeps them to ~1 in 2000."""
    return ((1, 6, 1), 64) if blur == "mild" else ((1, 1, 1), 9)


def canvas_np(k: int, height: int, width: int, blur=True, seed: int = SEED) -> np.ndarray:
    """uint8 texture canvas ``(height + 2*MARGIN) x (width + 2*MARGIN)`` for pair ``k``."""
    hh, ww = height + 2 * MARGIN, width + 2 * MARGIN
    pad = 1 if blur else 0
    y = np.arange(hh + 2 * pad, dtype=np.uint64)[:, None]
    x = np.arange(ww + 2 * pad, dtype=np.uint64)[None, :]
    t = (_mix32_np(seed, np.uint64(k) + np.zeros((1, 1), np.uint64), y, x) >> np.uint64(24)).astype(np.int32)
    if not blur:
        return t.astype(np.uint8)
    taps, div = _blur_taps(blur)
    acc = np.zeros((hh, ww), np.int32)
    for oy in range(3):
        for ox in range(3):
            acc += taps[oy] * taps[ox] * t[oy:oy + hh, ox:ox + ww]
    return ((acc + div // 2) // div).astype(np.uint8)


def pair_np(k: int, height: int, width: int, dx: int, dy: int, blur=True, seed: int = SEED,
            kind: str = "shift", noise: int = 3) -> tuple[np.ndarray, np.ndarray]:
    """(cur, prev) uint8 frames of pair ``k``.

    kind: "shift" planted integer translation; "identical" cur == prev;
    "constant" both frames one grey level; "noisy" shift plus uniform noise of
    +-``noise`` LSB on cur (clamped to 0..255).
    """
    m = MARGIN
    assert abs(dx) < m and abs(dy) < m
    if kind == "constant":
        level = np.uint8((k * 37 + 11) % 256)
        f = np.full((height, width), level, np.uint8)
        return f.copy(), f
    c = canvas_np(k, height, width, blur, seed)
    prev = c[m:m + height, m:m + width]
    if kind == "identical":
        return prev.copy(), prev.copy()
    cur = c[m - dy:m - dy + height, m - dx:m - dx + width].copy()
    if kind == "noisy":
        y = np.arange(height, dtype=np.uint64)[:, None]
        x = np.arange(width, dtype=np.uint64)[None, :]
        n = _mix32_np(seed ^ 0xA5A5A5A5, np.uint64(k) + np.zeros((1, 1), np.uint64), y, x)
        n = (n % np.uint64(2 * noise + 1)).astype(np.int32) - noise
        cur = np.clip(cur.astype(np.int32) + n, 0, 255).astype(np.uint8)
    return cur, prev.copy()


def batch_np(n_pairs: int, height: int, width: int, s: int, blur=True, seed: int = SEED,
             k0: int = 0, classes: bool = True):
    """Batch ``[n, H, W]`` uint8 cur/prev plus the planted shifts and class names.

    With ``classes`` every 20th pair is identical / constant / noisy (each ~5 %).
    """
    cur = np.empty((n_pairs, height, width), np.uint8)
    prev = np.empty((n_pairs, height, width), np.uint8)
    shifts = np.zeros((n_pairs, 2), np.int32)
    kinds = []
    for i in range(n_pairs):
        k = k0 + i
        dx, dy = planted_shift(k, s)
        kind = "shift"
        if classes:
            kind = {3: "identical", 7: "constant", 11: "noisy"}.get(k % 20, "shift")
        if kind in ("identical", "constant"):
            dx = dy = 0
        cur[i], prev[i] = pair_np(k, height, width, dx, dy, blur, seed, kind)
        shifts[i] = (dx, dy)
        kinds.append(kind)
    return cur, prev, shifts, kinds


# ----------------------------------------------------------------------------------------------
# torch twin (any device): identical bytes to batch_np(..., classes=...) for the same arguments.
# ----------------------------------------------------------------------------------------------

def _mix32_t(seed: int, k, y, x):
    import torch  # noqa: F401

    h = (seed ^ (k * _K_PAIR)) & _M32
    h = (h ^ (y * _K_ROW)) & _M32
    h = (h ^ (x * _K_COL)) & _M32
    h = h ^ (h >> 16)
    h = (h * _F1) & _M32
    h = h ^ (h >> 13)
    h = (h * _F2) & _M32
    h = h ^ (h >> 16)
    return h


def batch_torch(n_pairs: int, height: int, width: int, s: int, device, blur=True, seed: int = SEED,
                k0: int = 0, classes: bool = True, chunk: int = 32):
    """torch version of :func:`batch_np`; returns (cur, prev, shifts[n,2] int32 cpu tensor, kinds)."""
    import torch

    m = MARGIN
    cur = torch.empty((n_pairs, height, width), dtype=torch.uint8, device=device)
    prev = torch.empty((n_pairs, height, width), dtype=torch.uint8, device=device)
    shifts = torch.zeros((n_pairs, 2), dtype=torch.int32)
    kinds = []
    pad = 1 if blur else 0
    hh, ww = height + 2 * m, width + 2 * m
    y = torch.arange(hh + 2 * pad, dtype=torch.int64, device=device)[None, :, None]
    x = torch.arange(ww + 2 * pad, dtype=torch.int64, device=device)[None, None, :]
    yn = torch.arange(height, dtype=torch.int64, device=device)[:, None]
    xn = torch.arange(width, dtype=torch.int64, device=device)[None, :]
    for c0 in range(0, n_pairs, chunk):
        c1 = min(n_pairs, c0 + chunk)
        ks = torch.arange(k0 + c0, k0 + c1, dtype=torch.int64, device=device)[:, None, None]
        t = _mix32_t(seed, ks, y, x) >> 24
        if blur:
            taps, div = _blur_taps(blur)
            acc = torch.zeros((c1 - c0, hh, ww), dtype=torch.int64, device=device)
            for oy in range(3):
                for ox in range(3):
                    acc += taps[oy] * taps[ox] * t[:, oy:oy + hh, ox:ox + ww]
            canvas = ((acc + div // 2) // div).to(torch.uint8)
        else:
            canvas = t.to(torch.uint8)
        for i in range(c0, c1):
            k = k0 + i
            dx, dy = planted_shift(k, s)
            kind = "shift"
            if classes:
                kind = {3: "identical", 7: "constant", 11: "noisy"}.get(k % 20, "shift")
            if kind in ("identical", "constant"):
                dx = dy = 0
            cv = canvas[i - c0]
            if kind == "constant":
                level = (k * 37 + 11) % 256
                cur[i].fill_(level)
                prev[i].fill_(level)
            else:
                prev[i] = cv[m:m + height, m:m + width]
                cur[i] = cv[m - dy:m - dy + height, m - dx:m - dx + width]
                if kind == "noisy":
                    n = _mix32_t(seed ^ 0xA5A5A5A5, k, yn, xn)
                    n = (n % 7) - 3
                    cur[i] = torch.clamp(cur[i].to(torch.int64) + n, 0, 255).to(torch.uint8)
            shifts[i, 0], shifts[i, 1] = dx, dy
            kinds.append(kind)
    return cur, prev, shifts, kinds


def video_torch(n_frames: int, height: int, width: int, device, k: int = 0, blur=True, seed: int = SEED):
    """A synthetic VIDEO on `device`: frame t is the height x width window of one canvas (texture index k) at an offset that
    follows a closed Lissajous path inside the canvas margin, so consecutive frames are translated copies of each other
    (what the sequence entry points -- K1 on frames[1:] / frames[:-1], the estimator's sequence mode -- are meant for).
    Returns (frames uint8 [n_frames, height, width], offsets int32 [n_frames, 2] cpu (x, y))."""
    import math

    import torch

    m = MARGIN
    pad = 1 if blur else 0
    hh, ww = height + 2 * m, width + 2 * m
    y = torch.arange(hh + 2 * pad, dtype=torch.int64, device=device)[:, None]
    x = torch.arange(ww + 2 * pad, dtype=torch.int64, device=device)[None, :]
    t = _mix32_t(seed, k, y, x) >> 24
    if blur:
        taps, div = _blur_taps(blur)
        acc = torch.zeros((hh, ww), dtype=torch.int64, device=device)
        for oy in range(3):
            for ox in range(3):
                acc += taps[oy] * taps[ox] * t[oy:oy + hh, ox:ox + ww]
        canvas = ((acc + div // 2) // div).to(torch.uint8)
    else:
        canvas = t.to(torch.uint8)
    frames = torch.empty((n_frames, height, width), dtype=torch.uint8, device=device)
    offs = torch.zeros((n_frames, 2), dtype=torch.int32)
    for f in range(n_frames):
        ox = int(round((m - 1) * math.sin(2.0 * math.pi * f / 97.0)))
        oy = int(round((m - 1) * math.sin(2.0 * math.pi * f / 61.0 + 0.5)))
        frames[f] = canvas[m + oy:m + oy + height, m + ox:m + ox + width]
        offs[f, 0], offs[f, 1] = ox, oy
    return frames, offs


# ----------------------------------------------------------------------------------------------
# Input classes the random fuzzers found worth pinning (tools/fft_sr_fuzz.py, r03 / r04): every one of them once broke, or
# could break, a kernel that passes on plain shifted texture. Seeded, so a regression shows in the test suite.
# ----------------------------------------------------------------------------------------------

def smooth_np(k: int, height: int, width: int, passes: int = 4, seed: int = SEED) -> np.ndarray:
    """Strongly low-passed texture (``passes`` integer 3x3 box blurs): most cross-power bins sit near the f32 rounding floor --
    the content class on which f32 arithmetic, not the algorithm, limits agreement (Tolerances)."""
    c = canvas_np(k, height + 2 * passes, width + 2 * passes, False, seed).astype(np.int32)
    for _ in range(passes):
        hh, ww = c.shape[0] - 2, c.shape[1] - 2
        acc = np.zeros((hh, ww), np.int32)
        for oy in range(3):
            for ox in range(3):
                acc += c[oy:oy + hh, ox:ox + ww]
        c = (acc + 4) // 9
    m = MARGIN
    return c[m:m + height, m:m + width].astype(np.uint8)


def fuzz_classes_np(k: int, height: int, width: int, dx: int = 3, dy: int = -2, seed: int = SEED) -> dict:
    """name -> (cur, prev): the degenerate / ill-conditioned classes, all ``height x width`` uint8.
      const_cur / const_prev   exactly ONE frame of the pair constant (r03: the packed two-for-one transform answered with noise)
      black_cur / black_prev / black_both   all-zero frames (the only constant patch that survives zero padding; SR: the
                               all-zero log-polar image)
      const_rect               a constant rectangle inside a textured frame: some patches constant, their neighbours not
      saturated                a region clipped to 255 in both frames (constant where it covers a patch)
      smooth                   strongly low-passed content, shifted (f32-limited)
      checker                  texture whose alternating-sign pixel sums cancel in places (r04: the real-only CCS slots of small
                               patches must come out of the transform exactly)"""
    cur, prev = pair_np(k, height, width, dx, dy, True, seed)
    level = np.uint8((k * 37 + 11) % 255 + 1)
    const = np.full((height, width), level, np.uint8)
    black = np.zeros((height, width), np.uint8)
    out = {"const_cur": (const.copy(), prev.copy()), "const_prev": (cur.copy(), const.copy()),
           "black_cur": (black.copy(), prev.copy()), "black_prev": (cur.copy(), black.copy()), "black_both": (black.copy(), black.copy())}
    rc, rp = cur.copy(), prev.copy()
    y0, y1, x0, x1 = height // 5, (3 * height) // 4, width // 6, (2 * width) // 3
    rc[y0:y1, x0:x1] = level
    out["const_rect"] = (rc, rp)
    sc, sp = cur.copy(), prev.copy()
    sc[: height // 2, : (3 * width) // 5] = 255
    sp[: height // 2, : (3 * width) // 5] = 255
    out["saturated"] = (sc, sp)
    sm = smooth_np(k, height + 2 * MARGIN, width + 2 * MARGIN, 4, seed)
    m = MARGIN
    out["smooth"] = (sm[m - dy:m - dy + height, m - dx:m - dx + width].copy(), sm[m:m + height, m:m + width].copy())
    # two-level texture in 2 x 2 cells: many patches whose (-1)^x, (-1)^y or (-1)^(x+y) weighted pixel sums are exactly zero
    cell = (canvas_np(k + 1, height, width, False, seed)[MARGIN:MARGIN + height, MARGIN:MARGIN + width] >> 7).astype(np.uint8)
    cell = np.repeat(np.repeat(cell[::2, ::2], 2, axis=0), 2, axis=1)[:height, :width]
    ck = (cell * 200 + 20).astype(np.uint8)
    out["checker"] = (np.roll(ck, (2, 2), axis=(0, 1)), ck)
    return out

#!/usr/bin/env python3
"""Writes the A/B inputs and the oracle's answers as raw files + manifest.json (see README.md). Runs anywhere the
oracle builds; needs neither OpenCV nor a GPU."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import geom_scenes as S  # noqa: E402
import oracle_lib as O  # noqa: E402
import sr_scenes  # noqa: E402
from mrs_optic_flow_amd import synth  # noqa: E402

out = sys.argv[1] if len(sys.argv) > 1 else "ab_data"
os.makedirs(out, exist_ok=True)
cases = []


def put(name, arr):
    arr = np.ascontiguousarray(arr)
    arr.tofile(os.path.join(out, name + ".bin"))
    return {"file": name + ".bin", "dtype": str(arr.dtype), "shape": list(arr.shape)}


# ---- phase correlation: patches of the committed golden frames + circular shifts + a constant patch
g = np.load(os.path.join(ROOT, "tests", "golden", "fft_n64_unaligned.npz"))
w, h, n, gx, gy, ox, oy, sx, sy = (int(v) for v in g["layout"])
k = 0
for pair in range(min(3, g["cur"].shape[0])):
    for (i, j) in ((0, 0), (gx - 1, gy - 1), (gx // 2, gy // 2)):
        a = g["cur"][pair, oy + j * sy: oy + j * sy + n, ox + i * sx: ox + i * sx + n].astype(np.float32)
        b = g["prev"][pair, oy + j * sy: oy + j * sy + n, ox + i * sx: ox + i * sx + n].astype(np.float32)
        (x, y), _ = O.phase_correlate(a, b, 32)
        cases.append({"kind": "pc", "name": f"pc_{k}", "n": n, "a": put(f"pc_{k}_a", a), "b": put(f"pc_{k}_b", b),
                      "oracle": put(f"pc_{k}_oracle", np.array([x, y]))})
        k += 1
for n in (64, 120, 128, 480):
    base = synth.canvas_np(3 + n, n, n, True)[:n, :n].astype(np.float32)
    for (dx, dy) in ((5, -3), (0, 0)):
        a = np.roll(base, (dy, dx), axis=(0, 1))
        (x, y), _ = O.phase_correlate(a, base, 32)
        cases.append({"kind": "pc", "name": f"pc_{k}", "n": n, "a": put(f"pc_{k}_a", a), "b": put(f"pc_{k}_b", base),
                      "oracle": put(f"pc_{k}_oracle", np.array([x, y]))})
        k += 1
# r04: sizes without a tuned kernel -- 5-smooth even sizes, sizes cv::phaseCorrelate pads (62 -> 64, 98 -> 100), sizes it pads to an
# ODD transform (74 -> 75, 124 -> 125), odd sizes, a large patch -- on NON-circular shifts (real padding / windowing effects) and on
# identical images (an odd padded size answers (0.5, 0.5): the centre M / 2.0 against the shifted origin M >> 1)
for n in (60, 80, 96, 100, 62, 98, 74, 124, 75, 45, 160, 144):
    for (dx, dy) in ((3, -2), (0, 0)):
        cur, prev = synth.pair_np(500 + n, n, n, dx, dy, True)
        a, b = cur.astype(np.float32), prev.astype(np.float32)
        (x, y), _ = O.phase_correlate(a, b, 32)
        cases.append({"kind": "pc", "name": f"pc_{k}", "n": n, "a": put(f"pc_{k}_a", a), "b": put(f"pc_{k}_b", b),
                      "oracle": put(f"pc_{k}_oracle", np.array([x, y]))})
        k += 1
# cv::getOptimalDFTSize, which decides those paddings
sizes = np.array(list(range(1, 520)) + [959, 960, 961, 1000, 4097], np.int32)
cases.append({"kind": "optdft", "name": "optdft_0", "count": int(sizes.size), "sizes": put("optdft_0_sizes", sizes),
              "oracle": put("optdft_0_oracle", np.array([O.optimal_dft_size(int(v)) for v in sizes], np.int32))})
# one-sided constant patch on a padded size: the reference's box spectrum has exact Nyquist zeros (r04)
cur62 = np.full((62, 62), 197, np.float32)
prev62 = synth.canvas_np(9, 62, 62, True)[:62, :62].astype(np.float32)
(x, y), _ = O.phase_correlate(cur62, prev62, 32)
cases.append({"kind": "pc", "name": f"pc_{k}", "n": 62, "a": put(f"pc_{k}_a", cur62), "b": put(f"pc_{k}_b", prev62),
              "oracle": put(f"pc_{k}_oracle", np.array([x, y]))})
k += 1
const = np.full((64, 64), 200, np.float32)
(x, y), _ = O.phase_correlate(const, const, 32)
cases.append({"kind": "pc", "name": f"pc_{k}", "n": 64, "a": put(f"pc_{k}_a", const), "b": put(f"pc_{k}_b", const),
              "oracle": put(f"pc_{k}_oracle", np.array([x, y]))})

# ---- log-polar remap (both interpolations, both OpenCV generations) and the bare maps
k = 0
for res, M in ((240, 40.0), (256, 45.0), (480, 49.9), (320, 45.0), (350, 49.9)):  # (320 / 350: planned transforms, r04; 350 pads to 360)
    base = sr_scenes.canvas(5 + res, res)
    src = sr_scenes.view(base, res, 1.05, 7.0)
    src[:5] = 255
    entry = {"kind": "lp", "name": f"lp_{k}", "res": res, "M": M, "src": put(f"lp_{k}_src", src), "oracle": {}}
    for interp in (2, 4):
        for variant in (0, 1):
            dst = O.logpolar(src, M, interp, dst=np.full((res, res), 37, np.uint8), variant=variant)
            entry["oracle"][f"i{interp}_v{variant}"] = put(f"lp_{k}_oracle_i{interp}_v{variant}", dst)
    for variant in (0, 1):
        mx, my = O.logpolar_maps(res, M, variant)
        entry["oracle"][f"mapx_v{variant}"] = put(f"lp_{k}_mapx_v{variant}", mx)
        entry["oracle"][f"mapy_v{variant}"] = put(f"lp_{k}_mapy_v{variant}", my)
    cases.append(entry)
    k += 1

# ---- the estimator as a STREAM (scaleRotationEstimator.cpp:34-148): first frame INTER_CUBIC -> (1, 0); then INTER_LANCZOS4,
#      cv::phaseCorrelate(tempIm_F32, prevIm_F32), the gate of :119-121, prev <- cur. What the sequence entry
#      (mof_sr_process_sequence_device) and the stateful mof_sr_process reproduce.
for k, (res, M, nf) in enumerate(((240, 40.0, 6), (480, 49.9, 4), (350, 49.9, 4))):
    base = sr_scenes.canvas(31 + res, res)
    video = np.stack([sr_scenes.view(base, res, 1.0 + 0.015 * t, 1.7 * t) for t in range(nf)])
    ref = O.ScaleRotationEstimator(res, M, 32)
    want = []
    for t in range(nf):
        sc, ro = ref.processImage(video[t])
        want.append([sc, ro, ref.pt[0] if t else 0.0, ref.pt[1] if t else 0.0])
    cases.append({"kind": "srseq", "name": f"srseq_{k}", "res": res, "M": M, "frames": nf, "video": put(f"srseq_{k}_video", video),
                  "oracle": put(f"srseq_{k}_oracle", np.array(want))})

# ---- resize 1/4 and x2, RGB2GRAY
img = synth.canvas_np(77, 128, 160, True)[:128, :160].copy()
cases.append({"kind": "resize_quarter", "name": "rq_0", "src": put("rq_0_src", img), "oracle": put("rq_0_oracle", O.resize_quarter(img))})
cases.append({"kind": "resize_2x", "name": "r2_0", "src": put("r2_0_src", img), "oracle": put("r2_0_oracle", O.resize_2x(img))})
rng = np.random.default_rng(5)
col = rng.integers(0, 256, (64, 96, 3), dtype=np.uint8)
cases.append({"kind": "gray", "name": "gray_0", "src": put("gray_0_src", col), "oracle": put("gray_0_oracle", O.rgb2gray(col))})

# ---- geometry tail: undistortion, RANSAC homography, decomposition
cam = (340.0, 338.5, 376.0, 240.0, -0.28, 0.07, 0.0004, -0.0003, -0.006)
uv = np.stack([rng.uniform(0, 480, 200), rng.uniform(0, 480, 200)], axis=1)
cases.append({"kind": "undistort", "name": "und_0", "camera": list(cam), "ul_corner_x": 136.0, "pts": put("und_0_pts", uv),
              "oracle": put("und_0_oracle", O.geom_undistort(O.GeomCamera(*cam), 136.0, uv))})
for k in range(4):
    R = S.rot_axis_angle(rng.normal(size=3), rng.uniform(0.002, 0.05))
    H = S.plane_homography(R, rng.normal(0, 0.03, 3), np.array([0.02, -0.01, 1.0]), 1.0)
    a = rng.uniform(-0.7, 0.7, (64, 2))
    hh = np.concatenate([a, np.ones((64, 1))], axis=1) @ H.T
    b = hh[:, :2] / hh[:, 2:3]
    outl = rng.random(64) < 0.25
    outl[:4] = False
    b[outl] += rng.choice([-1.0, 1.0], (int(outl.sum()), 2)) * rng.uniform(0.03, 0.3, (int(outl.sum()), 2))
    Ho, mask = O.geom_find_homography(a, b)
    Rs, ts, ns = O.geom_decompose(Ho)
    cases.append({"kind": "homography", "name": f"hom_{k}", "a": put(f"hom_{k}_a", a), "b": put(f"hom_{k}_b", b),
                  "oracle": {"H": put(f"hom_{k}_oracle_H", Ho), "mask": put(f"hom_{k}_oracle_mask", mask),
                             "R": put(f"hom_{k}_oracle_R", Rs), "t": put(f"hom_{k}_oracle_t", ts), "n": put(f"hom_{k}_oracle_n", ns)}})

# plain-text twin of the manifest for the C++ side (no JSON parser needed there)
with open(os.path.join(out, "manifest.txt"), "w") as f:
    for c in cases:
        if c["kind"] == "pc":
            f.write(f"pc {c['name']} {c['n']}\n")
        elif c["kind"] == "lp":
            f.write(f"lp {c['name']} {c['res']} {c['M']!r}\n")
        elif c["kind"] in ("resize_quarter", "resize_2x", "gray"):
            f.write(f"{c['kind']} {c['name']} {c['src']['shape'][0]} {c['src']['shape'][1]}\n")
        elif c["kind"] == "undistort":
            f.write(f"undistort {c['name']} {c['pts']['shape'][0]} " + " ".join(repr(v) for v in c["camera"]) + f" {c['ul_corner_x']!r}\n")
        elif c["kind"] == "homography":
            f.write(f"homography {c['name']} {c['a']['shape'][0]}\n")
        elif c["kind"] == "optdft":
            f.write(f"optdft {c['name']} {c['count']}\n")
        elif c["kind"] == "srseq":
            f.write(f"srseq {c['name']} {c['res']} {c['M']!r} {c['frames']}\n")
json.dump({"cases": cases, "oracle_version": O.lib().oracle_version().decode()}, open(os.path.join(out, "manifest.json"), "w"), indent=1)
print(f"wrote {len(cases)} cases to {out}/")

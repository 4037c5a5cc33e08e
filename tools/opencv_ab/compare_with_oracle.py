#!/usr/bin/env python3
"""Diffs OpenCV's outputs (dump_opencv) with the oracle's (export_inputs.py). See README.md. Exit code 1 on a mismatch."""
import json
import os
import sys

import numpy as np

d = sys.argv[1] if len(sys.argv) > 1 else "ab_data"
man = json.load(open(os.path.join(d, "manifest.json")))
bad = 0


def rd(name, dtype, shape=None):
    p = os.path.join(d, name + ".bin")
    if not os.path.exists(p):
        return None
    a = np.fromfile(p, dtype=dtype)
    return a.reshape(shape) if shape is not None else a


def oracle(entry):
    return np.fromfile(os.path.join(d, entry["file"]), dtype=entry["dtype"]).reshape(entry["shape"])


def report(name, what, ok, detail):
    global bad
    print(f"{'ok  ' if ok else 'DIFF'} {name:10s} {what:34s} {detail}")
    bad += 0 if ok else 1


for c in man["cases"]:
    k, name = c["kind"], c["name"]
    if k == "pc":
        cv = rd(name + "_cv", np.float64)
        if cv is None:
            continue
        o = oracle(c["oracle"])
        err = float(np.abs(cv - o).max())
        report(name, "cv::phaseCorrelate", err <= 1e-4, f"cv {cv} oracle {o} |d| {err:.2e} (bar 1e-4 px)")
    elif k == "optdft":
        cv = rd(name + "_cv", np.int32)
        if cv is None:
            continue
        o = oracle(c["oracle"])
        n = int((cv != o).sum())
        report(name, "cv::getOptimalDFTSize", n == 0, f"{n} of {o.size} sizes differ")
    elif k == "lp":
        res = c["res"]
        for interp in (2, 4):
            for call, variant in (("logPolar", 0), ("cvLogPolar", 1)):
                cv = rd(f"{name}_cv_{call}_i{interp}", np.uint8, (res, res))
                if cv is None:
                    continue
                for v in (0, 1):  # which restated generation does this OpenCV's call follow?
                    o = oracle(c["oracle"][f"i{interp}_v{v}"])
                    n = int((cv != o).sum())
                    report(name, f"cv::{call} interp {interp} vs variant {v}", n == 0 if v == variant else True,
                           f"{n} of {res * res} bytes differ" + ("" if v == variant else " (informational)"))
            for v in (0, 1):
                cv = rd(f"{name}_cv_remap_i{interp}_v{v}", np.uint8, (res, res))
                if cv is not None:
                    n = int((cv != oracle(c["oracle"][f"i{interp}_v{v}"])).sum())
                    report(name, f"cv::remap on oracle maps i{interp} v{v}", n == 0, f"{n} bytes differ (remap fixed point only)")
    elif k == "srseq":
        o = oracle(c["oracle"])
        cv = rd(name + "_cv", np.float64, o.shape)
        if cv is not None:
            e_pt = float(np.abs(cv[:, 2:] - o[:, 2:]).max())
            e_sr = float(np.abs(cv[:, :2] - o[:, :2]).max())
            report(name, "scaleRotationEstimator stream (logPolar + phaseCorrelate per frame)", e_pt <= 1e-4 and e_sr <= 1e-5,
                   f"max |d pt| {e_pt:.2e} px (bar 1e-4), max |d (scale, rot)| {e_sr:.2e} (bar 1e-5)")
    elif k in ("resize_quarter", "resize_2x", "gray"):
        o = oracle(c["oracle"])
        cv = rd(name + "_cv", np.uint8, o.shape)
        if cv is not None:
            n = int((cv != o).sum())
            report(name, {"resize_quarter": "cv::resize 1/4", "resize_2x": "cv::resize x2", "gray": "cv::cvtColor RGB2GRAY"}[k],
                   n == 0, f"{n} of {o.size} bytes differ")
    elif k == "undistort":
        o = oracle(c["oracle"])
        cv = rd(name + "_cv", np.float64, o.shape)
        if cv is not None:
            err = float(np.abs(cv - o).max())
            report(name, "cv::undistortPoints", err <= 1e-12, f"max |d| {err:.2e} (bar 1e-12)")
    elif k == "homography":
        H = rd(name + "_cv_H", np.float64, (3, 3))
        if H is None:
            continue
        oH, om = oracle(c["oracle"]["H"]), oracle(c["oracle"]["mask"])
        m = rd(name + "_cv_mask", np.uint8)
        report(name, "findHomography consensus set", bool((m == om).all()), f"{int((m != om).sum())} of {m.size} flags differ")
        err = float(np.abs(H / H[2, 2] - oH).max())
        report(name, "findHomography H (own sampler!)", err <= 1e-6, f"max |d| {err:.2e} (bar 1e-6)")
        dec = rd(name + "_cv_decomp", np.float64)
        if dec is not None and dec.size % 15 == 0:
            dec = dec.reshape(-1, 15)
            # decompose the OPENCV homography with the oracle so that only the decomposition is compared
            sys.path[:0] = [os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests")]
            import oracle_lib as O
            Rs, ts, ns = O.geom_decompose(H)
            same = dec.shape[0] == Rs.shape[0]
            err = max((float(np.abs(dec[s, :9] - Rs[s].ravel()).max() + np.abs(dec[s, 9:12] - ts[s]).max() + np.abs(dec[s, 12:] - ns[s]).max())
                       for s in range(min(dec.shape[0], Rs.shape[0]))), default=0.0)
            report(name, "decomposeHomographyMat (order, signs)", same and err <= 1e-9, f"{dec.shape[0]} vs {Rs.shape[0]} solutions, max |d| {err:.2e}")
print(f"{bad} mismatching comparison(s)")
sys.exit(1 if bad else 0)

#!/usr/bin/env python3
"""Latency of the node's own call pattern: FftMethod.processImage(one host frame) -> host vector, reference default geometry
(480 x 480 crop, 4 x 4 patches of 120 x 120) and c2's. usage (GPU box): python tools/stateful_latency.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mrs_optic_flow_amd import FftMethod, synth

for name, kw, shape in (("ref 480^2, 4x4 x 120^2", dict(frame_size=480, sample_point_size=120, max_px_speed=80.0), (480, 480)),
                        ("c2 752x480, 8x8 x 64^2", dict(sample_point_size=64, frame_shape=(480, 752), grid=(8, 8), origin=(1, 1), stride=(98, 59)), (480, 752)),
                        # r04: a size without a tuned kernel (the planned kernel), and the reference's whole-frame fallback (470 / 100 does not
                        # divide: ONE 470 x 470 patch, padded to 480 -- the planned pipeline; 480 / 100: one 480 x 480 patch -- the estimator's kernels)
                        ("480^2, 8x8 x 60^2 (planned)", dict(frame_size=480, sample_point_size=60, max_px_speed=80.0), (480, 480)),
                        ("470^2 / 100 -> one 470^2 patch (planned-large)", dict(frame_size=470, sample_point_size=100, max_px_speed=80.0), (470, 470)),
                        ("480^2 / 100 -> one 480^2 patch (tuned transforms)", dict(frame_size=480, sample_point_size=100, max_px_speed=80.0), (480, 480))):
    fm = FftMethod(*([kw.pop("frame_size"), kw.pop("sample_point_size"), kw.pop("max_px_speed")] if "frame_size" in kw else []), **kw)
    frames = [synth.pair_np(5 + t, shape[0], shape[1], t % 5, -(t % 3))[0] for t in range(8)]
    for f in frames[:3]:
        fm.processImage(f)
    ts = []
    for t in range(200):
        f = frames[t % 8]
        t0 = time.perf_counter()
        fm.processImage(f)
        ts.append(time.perf_counter() - t0)
    ts = np.sort(np.array(ts)) * 1e3
    print(f"{name}: processImage(host frame) median {ts[100]:.3f} ms, p90 {ts[180]:.3f} ms, min {ts[0]:.3f} ms")

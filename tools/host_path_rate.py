#!/usr/bin/env python3
"""PCIe-inclusive rates of the host-buffer batch entries (mof_fft_process_batch_host / mof_bm_process_batch_host: host frames in, host
results out, csrc/host_pipe.hpp) -- never bench.py's `value` (that one starts with the batch resident in HBM); quoted in DESIGN.md section 6.
usage (GPU box): python tools/host_path_rate.py [pairs]            -> one line per memory layout, c2 and ref and c3"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mrs_optic_flow_amd import FastSpacedBMMethod, FftMethod, pinned_empty, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024


def rate(call, reps=4):
    call()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        call()
        best = min(best, time.perf_counter() - t0)
    return best


def frames_of(h, w):
    base, _, _, _ = synth.batch_np(64, h, w, 8, classes=False, k0=3)
    return np.ascontiguousarray(np.tile(base, ((n + 1 + 63) // 64, 1, 1))[: n + 1])


def table(name, eng, h, w):
    frames = frames_of(h, w)
    pin = pinned_empty(frames.shape)
    pin[:] = frames
    pc, pp = pinned_empty((n, h, w)), pinned_empty((n, h, w))
    pc[:] = frames[1:]
    pp[:] = frames[:-1]
    cur, prev = frames[1:].copy(), frames[:-1].copy()  # (two allocations: not a video)
    for label, c, p, frames_up in (("pageable pairs", cur, prev, 2 * n), ("pageable video", frames[1:], frames[:-1], n + 1),
                                   ("pinned pairs", pc, pp, 2 * n), ("pinned video", pin[1:], pin[:-1], n + 1)):
        t = rate(lambda: eng.process_batch_host(c, p))
        print(f"{name:5s} {label:15s}: {n / t:10,.0f} pairs/s  ({t * 1e3:7.1f} ms per {n} pairs, {frames_up * h * w / t / 1e9:5.1f} GB/s of frames over PCIe)", flush=True)


print(f"# host-pointer batch entries, {n} pairs per call, best of 4 (MOF_HOST_CHUNK={os.environ.get('MOF_HOST_CHUNK', 'default: 16 MB of frames')}, "
      f"MOF_HOST_THREADS={os.environ.get('MOF_HOST_THREADS', '4')})")
table("c2", FftMethod(sample_point_size=64, frame_shape=(480, 752), grid=(8, 8), origin=(1, 1), stride=(98, 59)), 480, 752)
table("ref", FftMethod(480, 120, 80.0), 480, 480)
table("c3", FastSpacedBMMethod(16, 16, 8, (480, 752)), 480, 752)

# the estimator's video from host memory (mof_sr_process_sequence_host): frames per second, every frame uploaded once
from mrs_optic_flow_amd import ScaleRotationEstimator
est = ScaleRotationEstimator(480, 49.9)
video = frames_of(480, 480)[:n]
pinv = pinned_empty(video.shape)
pinv[:] = video
for label, v in (("pageable video", video), ("pinned video", pinv)):
    t = rate(lambda: est.process_sequence_host(v))
    print(f"c5seq {label:15s}: {n / t:10,.0f} frames/s ({t * 1e3:7.1f} ms per {n} frames, {n * 480 * 480 / t / 1e9:5.1f} GB/s of frames over PCIe)", flush=True)

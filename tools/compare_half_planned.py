"""Rates of ONE patch size through the kernel family the environment selects (child processes set MOF_FFT_HALF / MOF_FFT_FORCE_PLANNED):
a 480 x 480 frame (496 for sizes that do not divide 480) tiled by n x n patches, 1024 frame pairs resident on the device, kernel time from
HIP events. usage: python tools/compare_half_planned.py <n> [<n> ...]   -> one line per size: n, variant, patch pairs / s, frame pairs / s"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mrs_optic_flow_amd import FftMethod, synth  # noqa: E402

dev = torch.device("cuda:0")
for n in [int(v) for v in sys.argv[1:]]:
    g = max(1, 480 // n)
    fs = g * n
    B = 1024 if n <= 128 else 512
    video, _ = synth.video_torch(B + 1, fs, fs, dev, k=n)
    fm = FftMethod(fs, n, 80.0)
    cur, prev = video[1:], video[:-1]
    for _ in range(5):
        fm.process_batch_device(cur, prev)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    steps = 30
    e0.record()
    for _ in range(steps):
        fm.process_batch_device(cur, prev)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    print(f"n={n} variant={fm.kernel_variant} patch_pairs_per_s={B * g * g / ms * 1e3:.0f} frame_pairs_per_s={B / ms * 1e3:.0f} ms={ms:.4f}", flush=True)
    del fm, video
    torch.cuda.empty_cache()

"""Video-mode rates of ONE patch size (mof_fft_process_sequence_device) through the kernel family the environment selects -- the sequence
twin of tools/compare_half_planned.py. usage: python tools/compare_half_planned_video.py <n> [<n> ...]"""
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
    for _ in range(5):
        fm.process_sequence_device(video)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    steps = 30
    e0.record()
    for _ in range(steps):
        fm.process_sequence_device(video)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    print(f"n={n} variant={fm.kernel_variant} video frame_pairs_per_s={B / ms * 1e3:.0f} ms={ms:.4f}")

#!/usr/bin/env python3
"""Pair entry on two independent batches against the video entry on the same number of pairs, across patch sizes: the video form should
never lose.  usage (GPU box): python tools/video_probe.py [n ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mrs_optic_flow_amd import FftMethod, synth

dev = torch.device("cuda", 0)
sizes = [int(v) for v in sys.argv[1:]] or [16, 24, 32, 40, 48, 50, 54, 60, 64, 72, 80, 90, 96, 100, 108, 120, 128, 136, 144, 150, 160, 180, 192, 200, 240, 250, 256, 300, 320, 400, 480, 512, 640, 720, 960]


def timed(f):
    f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(6):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 6


for n in sizes:
    g = max(1, min(8, 512 // n))
    side = g * n + 8
    B = max(8, min(512, (1 << 25) // (g * g * n * n)))
    video, _ = synth.video_torch(B + 1, side, side, dev, k=0)
    cur, prev, _, _ = synth.batch_torch(B, side, side, 6, dev, k0=0)
    fm = FftMethod(sample_point_size=n, frame_shape=(side, side), grid=(g, g), origin=(0, 0), stride=(n, n))
    tp = timed(lambda: fm.process_batch_device(cur, prev))
    tv = timed(lambda: fm.process_sequence_device(video))
    flag = "   <-- the video form loses" if tv > 1.03 * tp else ""
    print(f"n {n:4d}  grid {g}x{g}  {B:4d} pairs  pair entry {tp:8.3f} ms  video entry {tv:8.3f} ms  ratio {tp / tv:5.2f}  [{fm.kernel_variant}]{flag}", flush=True)
    del fm, video, cur, prev

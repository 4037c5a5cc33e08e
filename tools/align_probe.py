#!/usr/bin/env python3
"""Does the alignment of a patch's first pixel cost anything on the large-patch path? Same patch size, same frames, patch origins / strides
that put the second column of patches at x = 0, 1, 2, 3 (mod 4).  usage (GPU box): python tools/align_probe.py [n ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mrs_optic_flow_amd import FftMethod, synth

dev = torch.device("cuda", 0)
sizes = [int(v) for v in sys.argv[1:]] or [200, 250, 160, 120]
B = 256
for n in sizes:
    side = 2 * n + 8
    cur, prev, _, _ = synth.batch_torch(B, side, side, 6, dev, k0=0)
    for ox in (0, 1, 2, 3):
        fm = FftMethod(sample_point_size=n, frame_shape=(side, side), grid=(2, 2), origin=(ox, 0), stride=(n + (4 - n % 4) % 4, n))
        fm.process_batch_device(cur, prev)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fm.process_batch_device(cur, prev)
        e1.record()
        torch.cuda.synchronize()
        print(f"n {n:4d}  first pixel of every patch at x = {ox} (mod 4): {e0.elapsed_time(e1) / 10:8.3f} ms per {B} pairs", flush=True)

#!/usr/bin/env python3
"""Time per pixel of the large-patch path across transform sizes (2 x 2 patches, 256 pairs): a size that sticks out has a problem of its own.
usage (GPU box): python tools/size_probe.py [n ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mrs_optic_flow_amd import FftMethod, synth

dev = torch.device("cuda", 0)
sizes = [int(v) for v in sys.argv[1:]] or [200, 216, 240, 250, 256, 270, 288, 300, 320, 360, 384, 400, 432, 450, 480, 512]
for n in sizes:
    B = max(32, min(256, (1 << 26) // (4 * n * n)))
    side = 2 * n + 8
    cur, prev, _, _ = synth.batch_torch(B, side, side, 6, dev, k0=0)
    fm = FftMethod(sample_point_size=n, frame_shape=(side, side), grid=(2, 2), origin=(0, 0), stride=(n + (4 - n % 4) % 4, n))
    fm.process_batch_device(cur, prev)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fm.process_batch_device(cur, prev)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"n {n:4d}  {ms:8.3f} ms per {B} pairs  = {ms * 1e6 / (B * 8 * n * n):7.2f} ps per pixel", flush=True)
    del fm, cur, prev

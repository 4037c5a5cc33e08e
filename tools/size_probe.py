#!/usr/bin/env python3
"""Time per pixel of the FftMethod path across patch sizes (g x g patches tiling a ~512-pixel frame, ~64 Mpx per batch): a size that sticks
out from its neighbours has a problem of its own (r06: the sizes with N % 8 != 0 on the tuned large transforms -- Zh's row pitch).
usage (GPU box): python tools/size_probe.py [n ...]      (default: the tuned large sizes; `all`: every even size 16 .. 192 and a sample above)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mrs_optic_flow_amd import FftMethod, synth

dev = torch.device("cuda", 0)
args = sys.argv[1:]
if args == ["all"]:
    sizes = list(range(16, 193, 2)) + [193, 196, 199, 201, 210, 225, 243, 245, 280, 310, 324, 375, 405, 486, 500, 540, 600, 640, 720, 750, 810, 960]
else:
    sizes = [int(v) for v in args] or [200, 216, 240, 250, 256, 270, 288, 300, 320, 360, 384, 400, 432, 450, 480, 512]
for n in sizes:
    g = max(1, min(8, 512 // n))
    side = g * n + 8
    B = max(8, min(1024, (1 << 26) // (2 * g * g * n * n)))
    cur, prev, _, _ = synth.batch_torch(B, side, side, 6, dev, k0=0)
    fm = FftMethod(sample_point_size=n, frame_shape=(side, side), grid=(g, g), origin=(0, 0), stride=(n, n))
    fm.process_batch_device(cur, prev)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fm.process_batch_device(cur, prev)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    mpx = B * 2 * g * g * n * n / 1e6
    print(f"n {n:4d}  grid {g}x{g}  {ms:8.3f} ms per {B:4d} pairs  {ms * 1e3 / mpx:7.3f} us per Mpx  [{fm.kernel_variant}]", flush=True)
    del fm, cur, prev

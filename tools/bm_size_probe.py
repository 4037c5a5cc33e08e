#!/usr/bin/env python3
"""Byte-differences per second of the block-matching path across geometries (752 x 480 frames, 256 pairs): a geometry that sticks out from
its neighbours has a problem of its own.  usage (GPU box): python tools/bm_size_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mrs_optic_flow_amd import BlockMethod, FastSpacedBMMethod, synth

dev = torch.device("cuda", 0)
B, H, W = 256, 480, 752
cur, prev, _, _ = synth.batch_torch(B, H, W, 6, dev, k0=0)
rows = []
for sps in (4, 8, 12, 16, 24, 32):
    for r in (2, 4, 8, 12, 16):
        for step in (0, 4, 8, 16):
            try:
                bm = FastSpacedBMMethod(sps, r, step, (H, W))
            except Exception as e:
                continue
            nb = bm.cfg.grid_x * bm.cfg.grid_y
            if nb <= 0:
                continue
            bm.process_batch_device(cur, prev)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                bm.process_batch_device(cur, prev)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            diffs = B * nb * (2 * r + 1) ** 2 * sps * sps
            print(f"block {sps:2d} radius {r:2d} step {step:2d}: {nb:5d} blocks  {ms:8.3f} ms  {diffs / ms / 1e9:8.1f} T byte-differences/s", flush=True)
            del bm

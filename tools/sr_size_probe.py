#!/usr/bin/env python3
"""Frames per second of the estimator's video entry across resolutions (tuned transforms vs the planned pipeline: MOF_SR_TUNED_ALL=0 / 1).
usage (GPU box): python tools/sr_size_probe.py [res ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mrs_optic_flow_amd import ScaleRotationEstimator, synth

dev = torch.device("cuda", 0)
for res in [int(v) for v in sys.argv[1:]] or [200, 320, 360, 480, 500, 512, 640, 720, 960]:
    n = max(32, min(512, (1 << 27) // (res * res)))
    video, _ = synth.video_torch(n, res, res, dev, k=0)
    est = ScaleRotationEstimator(res, 49.9)
    est.process_sequence_device(video, resolve_gate=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        est.process_sequence_device(video, resolve_gate=False)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"res {res:4d}  {n:4d} frames  {ms:8.3f} ms  {n / ms * 1e3:10,.0f} frames/s", flush=True)
    del est, video

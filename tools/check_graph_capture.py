#!/usr/bin/env python3
"""Small-scale check that a multi-pass scale/rotation batch can be captured into a HIP graph and replayed
(batch_chunk = 2 so that 7 pairs take four pipeline passes; both lane settings, the two-lane form forks the
engine's second stream into the capture). Prints 'graph ok' or raises."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch

import sr_scenes
from mrs_optic_flow_amd import ScaleRotationEstimator

res = int(sys.argv[1]) if len(sys.argv) > 1 else 240
n_pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 7
base = sr_scenes.canvas(3, res)
views = np.stack([sr_scenes.view(base, res, 1.0 + 0.01 * k, 1.5 * k) for k in range(8)])
idx = np.arange(n_pairs) % 7
cur = torch.from_numpy(views[1:][idx]).cuda()
prev = torch.from_numpy(views[:-1][idx]).cuda()
for lanes in (1, 2):
    est = ScaleRotationEstimator(res, 45.0, batch_chunk=2, pipeline_lanes=lanes)
    want = est.process_batch_device(cur, prev).clone()   # eager
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            out = est.process_batch_device(cur, prev)    # captured
    out.zero_()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, want), (lanes, out, want)
    del g, est
print("graph ok", res, n_pairs)

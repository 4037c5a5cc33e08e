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
# a fresh engine holds one pair's worth of scratch: capturing its first batch must fail loudly (allocation is not
# capturable), and work after reserve()
from mrs_optic_flow_amd import MofError
fresh = ScaleRotationEstimator(res, 45.0, batch_chunk=4)
g0 = torch.cuda.CUDAGraph()
s0 = torch.cuda.Stream()
refused = False
with torch.cuda.stream(s0):
    try:
        with torch.cuda.graph(g0, stream=s0):
            fresh.process_batch_device(cur, prev)
    except (MofError, RuntimeError) as exc:
        refused = "reserve" in str(exc) or "capture" in str(exc)
assert refused, "capturing the first batch of a fresh engine must be refused"
torch.cuda.synchronize()
del g0, fresh
fresh = ScaleRotationEstimator(res, 45.0, batch_chunk=4)
fresh.reserve(n_pairs)
g1 = torch.cuda.CUDAGraph()
s1 = torch.cuda.Stream()
with torch.cuda.stream(s1):
    with torch.cuda.graph(g1, stream=s1):
        out1 = fresh.process_batch_device(cur, prev)
g1.replay()
torch.cuda.synchronize()
want1 = ScaleRotationEstimator(res, 45.0).process_batch_device(cur, prev)
torch.cuda.synchronize()
assert torch.equal(out1, want1)
del g1, fresh

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
# reserve(small), capture, then a LARGER eager batch: growing would move the scratch under the captured graph, so it must be
# refused (MOF_ERR_BUSY) while the engine is pinned, and the replay must still be right; release_captured() lifts the pin
from mrs_optic_flow_amd import release_captured
est = ScaleRotationEstimator(res, 45.0)
est.reserve(n_pairs)
want = est.process_batch_device(cur, prev).clone()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    with torch.cuda.graph(g, stream=side):
        out = est.process_batch_device(cur, prev)
big_c, big_p = cur.repeat(16, 1, 1), prev.repeat(16, 1, 1)
try:
    est.process_batch_device(big_c, big_p)
    raise SystemExit("a batch that grows the pinned scratch was accepted")
except MofError as exc:
    assert exc.code == -2 and "graph" in str(exc), exc
out.zero_()
g.replay()
torch.cuda.synchronize()
assert torch.equal(out, want)
del g
assert release_captured(est) == 1
grown = est.process_batch_device(big_c, big_p)
torch.cuda.synchronize()
assert torch.equal(grown[:n_pairs], want)
print("graph ok", res, n_pairs)

#!/usr/bin/env python3
"""Diagnostic: the c5 bench step (K1 on whole frames + scale/rotation on the 480^2 crop + torch.cat) captured into a
HIP graph, piece by piece. usage: check_graph_c5.py <mode> [n_pairs]   mode = crop | full"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

import torch


from mrs_optic_flow_amd import FftMethod, ScaleRotationEstimator, synth

mode = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
dev = torch.device("cuda", 0)
cur, prev, _, _ = synth.batch_torch(B, 480, 752, 8, dev, k0=0)
sr = ScaleRotationEstimator(480, 49.9)
cur_c, prev_c = cur[:, :480, 136:616], prev[:, :480, 136:616]
eng = FftMethod(sample_point_size=64, frame_shape=(480, 752), grid=(8, 8), origin=(1, 1), stride=(98, 59))
out = torch.empty((B, eng.n_patches, 2), dtype=torch.float64, device=dev)


def launch():
    if mode == "full":
        eng.process_batch_device(cur, prev, out=out)
    srout = sr.process_batch_device(cur_c, prev_c)
    if mode == "full":
        return torch.cat([out.reshape(B, -1), srout], dim=1)
    return srout


want = launch().clone()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    with torch.cuda.graph(g, stream=side):
        res = launch()
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
assert torch.equal(res, want)
print("graph ok", mode, B)

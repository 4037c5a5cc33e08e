#!/usr/bin/env python3
"""Stage cycle counts of the batched getRT kernel (library built with -DMOF_GEOM_PROF: the kernel then writes s_memtime
deltas instead of results: undistort + compaction, RANSAC, refit (DLT + LM), decomposition + pick). Diagnostic only."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch

from mrs_optic_flow_amd import FftMethod, geometry as G, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
cur, prev, _, _ = synth.batch_torch(B, 480, 480, 15, torch.device("cuda"), k0=0)
fm = FftMethod(480, 120, 80.0)
flow = fm.process_batch_device(cur, prev)
gcam = G.Camera(400.0, 400.0, 240.0, 240.0, -0.01, 0.002, 0.0, 0.0, 0.0)
gl = G.reference_layout(480, 120)
ident = (C.c_double * 4)(0, 0, 0, 1)
par = G.RtParams(3.0, 0.02, 0.0, ident, ident, (C.c_double * 3)(0, 0, 0))
row = np.frombuffer(bytes(par), dtype=np.float64).copy()
d_par = torch.from_numpy(np.repeat(row[None, :], B, axis=0)).cuda()
out = G.get_rt_batch_device(flow, gl, gcam, d_par, 8)
torch.cuda.synchronize()
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(10):
    out = G.get_rt_batch_device(flow, gl, gcam, d_par, 8)
t1.record(); torch.cuda.synchronize()
o = out.cpu().numpy()
print("kernel ms per batch of", B, ":", t0.elapsed_time(t1) / 10)
print("status histogram:", np.unique(o[:, 7], return_counts=True))
print("mean of out[:, :7] (stage cycles in a MOF_GEOM_PROF build):", o[:, :7].mean(axis=0).round(0))

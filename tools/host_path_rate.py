#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry point (mof_fft_process_batch: pageable host frames in, host results out) at c2:
never bench.py's `value` (that one starts with the batch resident in HBM); quoted in DESIGN.md section 6.
usage (GPU box): python tools/host_path_rate.py [pairs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mrs_optic_flow_amd import FftMethod, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
cur, prev, _, _ = synth.batch_np(64, 480, 752, 8, classes=False, k0=3)
cur = np.ascontiguousarray(np.tile(cur, (n // 64, 1, 1)))
prev = np.ascontiguousarray(np.tile(prev, (n // 64, 1, 1)))
fm = FftMethod(sample_point_size=64, frame_shape=(480, 752), grid=(8, 8), origin=(1, 1), stride=(98, 59))
fm.process_batch_host(cur[:8], prev[:8])
best = 1e9
for _ in range(3):
    t0 = time.perf_counter()
    out = fm.process_batch_host(cur, prev)
    best = min(best, time.perf_counter() - t0)
mb = (cur.nbytes + prev.nbytes) / 1e6
print(f"host path, c2, {n} pairs: {n / best:,.0f} pairs/s ({best * 1e3:.1f} ms per batch, {mb / best / 1e3:.1f} GB/s of frames over PCIe, pageable memory)")

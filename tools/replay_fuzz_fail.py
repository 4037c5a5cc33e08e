"""Replays the frame pairs tools/fft_sr_fuzz.py dumped on a mismatch (gpurun_out/fuzz_fail_*.npz) through the engine and the oracle.
usage: python tools/replay_fuzz_fail.py gpurun_out/fuzz_fail_*.npz"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import oracle_lib as O  # noqa: E402
from mrs_optic_flow_amd import FftMethod  # noqa: E402

dev = torch.device("cuda:0")
for f in sys.argv[1:]:
    d = np.load(f)
    cur, prev, n = d["cur"], d["prev"], int(d["n"])
    (gx, gy), (ox, oy), (sx, sy) = [tuple(int(v) for v in d[k]) for k in ("grid", "origin", "stride")]
    h, w = cur.shape
    fm = FftMethod(sample_point_size=n, frame_shape=(h, w), grid=(gx, gy), origin=(ox, oy), stride=(sx, sy))
    got = fm.process_batch_device(torch.from_numpy(cur[None]).to(dev), torch.from_numpy(prev[None]).to(dev)).cpu().numpy()[0]
    lay = O.fft_layout(w, h, n, gx, gy, (ox, oy), (sx, sy))
    w64, _ = O.fft_process(cur, prev, lay, 64)
    w32, _ = O.fft_process(cur, prev, lay, 32)
    e64, e32 = np.nanmax(np.abs(got - w64), initial=0.0), np.nanmax(np.abs(got - w32), initial=0.0)
    print(f"{os.path.basename(f)} n={n} variant={fm.kernel_variant} max|got-f64|={e64:.2e} max|got-f32|={e32:.2e} max|f32-f64|={np.nanmax(np.abs(w32 - w64), initial=0.0):.2e}")

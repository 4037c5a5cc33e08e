"""Quick on-box parity sweep of the fused half-tile kernel (csrc/pc_half_kernel.hip) against the oracle -- the development check behind
tests/test_gpu_half_tile.py. usage: MOF_FFT_HALF=1 python tools/check_half.py [sizes...]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
from mrs_optic_flow_amd import FftMethod, synth  # noqa: E402

sizes = [int(v) for v in sys.argv[1:]] or [64, 96, 120, 128, 144, 150, 160, 162, 180, 192, 140, 146, 158, 161, 170, 186, 137]
gpu = torch.device("cuda:0")
bad = 0
for n in sizes:
    gx, gy = 2, 2
    stride = (n + 3, n + 1)
    w, h = 5 + stride[0] * (gx - 1) + n + 2, 3 + stride[1] * (gy - 1) + n + 1
    B = 6
    cur, prev, shifts, kinds = synth.batch_np(B, h, w, max(1, n // 8), k0=n)
    fm = FftMethod(sample_point_size=n, frame_shape=(h, w), grid=(gx, gy), origin=(5, 3), stride=stride)
    got = fm.process_batch_device(torch.from_numpy(cur).to(gpu), torch.from_numpy(prev).to(gpu)).cpu().numpy()
    lay = O.fft_layout(w, h, n, gx, gy, (5, 3), stride)
    worst, nchk = 0.0, 0
    for k in range(B):
        want64, _, diags = O.fft_process(cur[k], prev[k], lay, 64, want_diag=True)
        for p in range(want64.shape[0]):
            if np.isnan(want64[p]).any():
                if not np.isnan(got[k][p]).all():
                    print("  NaN mismatch", n, k, p, got[k][p], want64[p]); bad += 1
                continue
            if not diags[p].second_value < 0.5 * diags[p].peak_value:
                continue
            d = float(np.abs(got[k][p] - want64[p]).max())
            worst = max(worst, d); nchk += 1
            if not d <= 1e-4:
                print("  off", n, k, kinds[k], p, got[k][p], want64[p], d); bad += 1
    print(f"n={n} variant={fm.kernel_variant} checked={nchk} worst={worst:.2e}")
print("BAD" if bad else "OK", bad)
sys.exit(1 if bad else 0)

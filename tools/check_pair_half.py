import os, sys
import numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import oracle_lib as O
from mrs_optic_flow_amd import FftMethod, synth
gpu = torch.device("cuda:0")
n = 128; gx, gy = 3, 2; stride = (n + 3, n - 9)
w, h = 5 + stride[0] * (gx - 1) + n + 2, 3 + stride[1] * (gy - 1) + n + 1
B = 150
cur, prev, shifts, kinds = synth.batch_np(B, h, w, 12, k0=7)
fm = FftMethod(sample_point_size=n, frame_shape=(h, w), grid=(gx, gy), origin=(5, 3), stride=stride)
got = fm.process_batch_device(torch.from_numpy(cur).to(gpu), torch.from_numpy(prev).to(gpu)).cpu().numpy()
lay = O.fft_layout(w, h, n, gx, gy, (5, 3), stride)
worst = 0; bad = 0; nchk = 0
for k in range(0, B, 7):
    want64, _, diags = O.fft_process(cur[k], prev[k], lay, 64, want_diag=True)
    for p in range(want64.shape[0]):
        if np.isnan(want64[p]).any():
            bad += not np.isnan(got[k][p]).all(); continue
        if not diags[p].second_value < 0.5 * diags[p].peak_value: continue
        d = float(np.abs(got[k][p] - want64[p]).max()); worst = max(worst, d); nchk += 1
        if d > 1e-4: bad += 1; print("off", k, kinds[k], p, got[k][p], want64[p])
print("pair-half check:", fm.kernel_variant, "checked", nchk, "worst", worst, "bad", bad)

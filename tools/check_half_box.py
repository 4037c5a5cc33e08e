"""On-box check of constant boxes through the half-tile kernel (csrc/pc_half_kernel.hip): ONE frame of a pair constant, patch size n
below its transform size M (the class the fuzzer's exceedances come from; profiles/r05_half_box_closed_ab.txt is its record). Prints, per case, the kernel's distance from the f64 oracle, from the f32 oracle, and the two
oracles' distance from each other, over the patches with a stable arg-max. usage: python tools/check_half_box.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
from mrs_optic_flow_amd import FftMethod, synth  # noqa: E402

gpu = torch.device("cuda:0")
cases = [(142, 81), (156, 160), (152, 212), (58, 97), (118, 120), (119, 7), (97, 200), (93, 33), (146, 84), (170, 255), (186, 1), (141, 90), (59, 128)]
bad = 0
for n, level in cases:
    for which in (0, 1):
        gx, gy = 2, 2
        stride = (n // 2 + 3, n // 3 + 1)
        w, h = 5 + stride[0] * (gx - 1) + n + 2, 3 + stride[1] * (gy - 1) + n + 1
        video, _ = synth.video_torch(2, h, w, "cpu", k=n + which)
        video[which] = level
        frames = video.numpy()
        fm = FftMethod(sample_point_size=n, frame_shape=(h, w), grid=(gx, gy), origin=(5, 3), stride=stride)
        dv = video.to(gpu)
        got = fm.process_batch_device(dv[1:], dv[:-1]).cpu().numpy()[0]
        lay = O.fft_layout(w, h, n, gx, gy, (5, 3), stride)
        want64, _, diags = O.fft_process(frames[1], frames[0], lay, 64, want_diag=True)
        want32, _ = O.fft_process(frames[1], frames[0], lay, 32)
        e64 = e32 = dd = 0.0
        nchk = 0
        for p in range(want64.shape[0]):
            if not diags[p].second_value < 0.5 * diags[p].peak_value:
                continue
            nchk += 1
            e64 = max(e64, float(np.abs(got[p] - want64[p]).max()))
            e32 = max(e32, float(np.abs(got[p] - want32[p]).max()))
            dd = max(dd, float(np.abs(want64[p] - want32[p]).max()))
        flag = "" if e64 <= 1e-4 else "  <-- above 1e-4 from f64"
        bad += e64 > 1e-4
        print(f"n={n} M={O.optimal_dft_size(n)} const={'prev' if which == 0 else 'cur'} level={level} {fm.kernel_variant} stable={nchk} "
              f"kernel-f64 {e64:.2e} kernel-f32 {e32:.2e} f64-f32 {dd:.2e}{flag}")
print("ABOVE", bad)

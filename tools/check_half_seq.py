"""On-box check of the half-tile kernel's VIDEO form (csrc/pc_half_kernel.hip, SEQ; mof_fft_process_sequence_device on the sizes that
kernel serves): videos with moving texture, a constant frame, a black frame and a repeated frame, longer than one run of 16 pairs and not a
multiple of it, patch sizes with and without padding -- every pair against the oracle (tests/tolerances.py) and against the pair entry
on the same frames. usage: python tools/check_half_seq.py [sizes...]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
import tolerances  # noqa: E402
from mrs_optic_flow_amd import FftMethod, synth  # noqa: E402

sizes = [int(v) for v in sys.argv[1:]] or [120, 60, 96, 100, 160, 144, 192, 118, 58, 137, 150, 180]
gpu = torch.device("cuda:0")
bad = 0
for n in sizes:
    gx, gy = 2, 2
    stride = (n // 2 + 3, n // 3 + 1)
    w, h = 5 + stride[0] * (gx - 1) + n + 2, 3 + stride[1] * (gy - 1) + n + 1
    F = 38  # frames: 37 pairs = two runs of 16 and one of 5
    video, _ = synth.video_torch(F, h, w, "cpu", k=n)
    video[7] = 93       # a constant frame: pairs 6 and 7 are degenerate / constant boxes
    video[20] = 0       # a black frame
    video[30] = video[29]  # a repeated frame: zero shift
    frames = video.numpy()
    fm = FftMethod(sample_point_size=n, frame_shape=(h, w), grid=(gx, gy), origin=(5, 3), stride=stride)
    dv = video.to(gpu)
    seq = fm.process_sequence_device(dv).cpu().numpy()
    pair = fm.process_batch_device(dv[1:], dv[:-1]).cpu().numpy()
    lay = O.fft_layout(w, h, n, gx, gy, (5, 3), stride)
    worst, nchk, same = 0.0, 0, 0
    for k in range(F - 1):
        want64, _, diags = O.fft_process(frames[k + 1], frames[k], lay, 64, want_diag=True)
        want32, _ = O.fft_process(frames[k + 1], frames[k], lay, 32)
        for p in range(want64.shape[0]):
            same += bool(np.array_equal(seq[k][p], pair[k][p], equal_nan=True))
            if np.isnan(want64[p]).any():
                if not np.isnan(seq[k][p]).all():
                    print("  NaN mismatch", n, k, p, seq[k][p], want64[p]); bad += 1
                continue
            if not diags[p].second_value < 0.5 * diags[p].peak_value:
                continue
            try:
                tolerances.check_patch(seq[k][p], want64[p], want32[p], f"halfseq{n}/pair{k}", p, pixels=tolerances.patch_pixels(frames[k + 1], frames[k], lay, p))
            except AssertionError as e:
                print("  off", n, k, p, seq[k][p], pair[k][p], want64[p], str(e)[:120]); bad += 1
            worst = max(worst, float(np.abs(seq[k][p] - want64[p]).max())); nchk += 1
    print(f"n={n} variant={fm.kernel_variant} checked={nchk} worst|seq-f64|={worst:.2e} results equal to the pair entry's bits: {same}/{(F - 1) * gx * gy}")
print("BAD" if bad else "OK", bad)
sys.exit(1 if bad else 0)

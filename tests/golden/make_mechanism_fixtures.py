"""Writes tests/golden/f32_mechanism_*.npz: ONE seeded frame pair per mechanism by which f32 arithmetic stops determining
-cv::phaseCorrelate's sub-pixel answer (tests/conditioning.py), with both oracles' answers.

  cancellation_const_vs_texture_n142   the pair tools/fft_sr_fuzz.py dumped in r05 (seed 605, sequence trial 14: a constant frame against texture, patch
                          142 on the 144 transform; profiles/r05_fuzz.txt) -- the inputs are read from that dump (gpurun_out/, scratch) when
                          it is present, otherwise regenerated from the fuzzer's own recipe is not possible (the trial's frames depend on
                          the whole random stream), so the committed .npz IS the record.
  exact_zero_bin_n48 / _n60   the two patches VERDICT r04 / r05 cited: the reference's own 480-px tiling, 3 x 3 box-blurred texture
                          (synth.pair_np(40 + n, 480, 480, 2t, -t)): prev has two bins that are zero in exact arithmetic.
  cancellation_smooth_n62 strongly low-passed content (synth.fuzz_classes_np(...)["smooth"]) on a padded size: the zero padding's edges dominate
                          the surface, the window's sum cancels 61-fold. (A scan of smooth content over sizes and blur strengths found bins under
                          the f32 rounding floor on 2 of 72 patches, with no measurable effect: "rounding-floor bin" stays a label conditioning.py
                          can give, but no fixture shows it as the dominant mechanism.)
Nothing here reads /root/reference.   Run:  python tests/golden/make_mechanism_fixtures.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.dirname(HERE), ROOT]

import oracle_lib as O  # noqa: E402
from mrs_optic_flow_amd import synth  # noqa: E402


def save(name, cur, prev, n, grid, origin, stride, patch, note):
    h, w = cur.shape
    lay = O.fft_layout(w, h, n, grid[0], grid[1], origin, stride)
    w64, _, diags = O.fft_process(cur, prev, lay, 64, want_diag=True)
    w32, _ = O.fft_process(cur, prev, lay, 32)
    np.savez_compressed(os.path.join(HERE, name), cur=cur, prev=prev, n=n, grid=np.array(grid), origin=np.array(origin), stride=np.array(stride),
                        patch=patch, oracle64=w64, oracle32=w32, stable=np.array([d.second_value < 0.5 * d.peak_value for d in diags]), note=note)
    print(name, cur.shape, "patch", patch, "o64", w64[patch], "o32-o64", np.abs(w32[patch] - w64[patch]).max())


if __name__ == "__main__":
    dump = os.path.join(ROOT, "gpurun_out", "fuzz_fail_seq_605_14_0.npz")
    if os.path.exists(dump):
        d = np.load(dump)
        save("f32_mechanism_cancellation_const_vs_texture_n142.npz", d["cur"], d["prev"], int(d["n"]), tuple(int(v) for v in d["grid"]),
             tuple(int(v) for v in d["origin"]), tuple(int(v) for v in d["stride"]), 0,
             "tools/fft_sr_fuzz.py 605 160 12, sequence trial 14, pair 0: constant frame (81) against texture, n = 142 -> M = 144")
    else:
        print("gpurun_out/fuzz_fail_seq_605_14_0.npz is gone: keeping the committed f32_mechanism_cancellation_const_vs_texture_n142.npz")
    for n, t, p in ((48, 1, 89), (60, 2, 22)):
        fs = 480
        seq = [synth.pair_np(40 + n, fs, fs, 2 * tt, -tt, blur=True)[0] for tt in range(3)]
        sq = fs // n
        i, j = p % sq, p // sq
        # a 2 x 1 crop of the tiling around the patch keeps the fixture small: the patch and its right (or left) neighbour
        i0 = min(i, sq - 2)
        cur, prev = seq[t][j * n:(j + 1) * n, i0 * n:(i0 + 2) * n], seq[t - 1][j * n:(j + 1) * n, i0 * n:(i0 + 2) * n]
        save(f"f32_mechanism_exact_zero_bin_n{n}.npz", np.ascontiguousarray(cur), np.ascontiguousarray(prev), n, (2, 1), (0, 0), (n, n), i - i0,
             f"reference tiling fs 480 / n {n}, frame t = {t}, patch {p} (VERDICT r05): synth.pair_np({40 + n}, 480, 480, 2t, -t, blur=True)")
    n = 62
    cur, prev = synth.fuzz_classes_np(n, n, n)["smooth"]
    save("f32_mechanism_cancellation_smooth_n62.npz", cur, prev, n, (1, 1), (0, 0), (n, n), 0, "synth.fuzz_classes_np(62, 62, 62)['smooth']: four 3 x 3 box blurs, 62 -> M = 64")

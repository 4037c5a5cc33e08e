"""Writes the golden fixtures in this directory.

The reference holds no fixtures, tests or images and cannot be built here
(OpenCV/ROS absent), so these vectors come from THIS repo's CPU oracle (oracle/*.c, fp64
variant for the FFT path; exact integers for block matching) on seeded synthetic frames.
They pin the oracle against regressions and give the GPU tests inputs + expected outputs
that do not depend on the oracle being importable. Parity with the real reference stays
"unpinned" (oracle/oracle.h). Nothing here reads /root/reference.

Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import oracle_lib as O  # noqa: E402
from mrs_optic_flow_amd import synth  # noqa: E402


def fft_case(name, h, w, n, grid, origin, stride, n_pairs, s, k0):
    cur, prev, shifts, kinds = synth.batch_np(n_pairs, h, w, s, k0=k0)
    lay = O.fft_layout(w, h, n, grid[0], grid[1], origin, stride)
    exp = np.zeros((n_pairs, grid[0] * grid[1], 2))
    well = np.zeros((n_pairs, grid[0] * grid[1]), bool)
    for k in range(n_pairs):
        exp[k], _, diags = O.fft_process(cur[k], prev[k], lay, 64, want_diag=True)
        # well-conditioned: a clear single peak (second-highest value outside the 5x5 window
        # below half the peak) -- only there is a 1e-4 px comparison meaningful
        well[k] = [d.second_value < 0.5 * d.peak_value for d in diags]
    np.savez_compressed(os.path.join(HERE, name), cur=cur, prev=prev, expected=exp, well_conditioned=well,
                        layout=np.array([w, h, n, grid[0], grid[1], origin[0], origin[1], stride[0], stride[1]]),
                        max_px_speed=80.0, planted=shifts, kinds=np.array(kinds))
    print(name, "pairs", n_pairs, "well-conditioned", int(well.sum()), "/", well.size)


def bm_case(name, h, w, block, step, radius, fast_spaced, n_pairs, s, k0):
    cur, prev, shifts, kinds = synth.batch_np(n_pairs, h, w, s, k0=k0)
    cfg = O.bm_config_fast_spaced(w, h, block, step, radius) if fast_spaced else O.bm_config_block_method(h, block, radius)
    dx = np.zeros((n_pairs, cfg.grid_y, cfg.grid_x), np.int8)
    dy = np.zeros_like(dx)
    mode = np.zeros((n_pairs, 2), np.int8)
    top = np.zeros((n_pairs, 2, 3), np.int8)
    for k in range(n_pairs):
        dx[k], dy[k], m = O.bm_process(cur[k], prev[k], cfg)
        mode[k] = m
        top[k, 0] = O.bm_histogram_top(dx[k], radius, 3)
        top[k, 1] = O.bm_histogram_top(dy[k], radius, 3)
    np.savez_compressed(os.path.join(HERE, name), cur=cur, prev=prev, dx=dx, dy=dy, mode=mode, top=top,
                        params=np.array([block, step, radius, int(fast_spaced)]), planted=shifts, kinds=np.array(kinds))
    print(name, "pairs", n_pairs, "grid", (cfg.grid_x, cfg.grid_y))


if __name__ == "__main__":
    # pairs k0.. chosen so that every class (shift / identical / constant / noisy) appears
    fft_case("fft_n64_unaligned.npz", 160, 224, 64, (3, 2), (1, 1), (79, 47), 12, 8, 0)
    fft_case("fft_n128.npz", 144, 272, 128, (2, 1), (5, 9), (139, 1), 12, 16, 0)
    fft_case("fft_n32_tiled.npz", 96, 96, 32, (3, 3), (0, 0), (32, 32), 12, 4, 0)
    # the reference's default patch size (config/default.yaml:32): 120 = 15 x 8, on a 240^2 crop -> 2 x 2 patches
    fft_case("fft_n120_reference_tiling.npz", 240, 240, 120, (2, 2), (0, 0), (120, 120), 8, 15, 0)
    bm_case("bm_fast_spaced_c3.npz", 120, 168, 16, 8, 16, True, 12, 12, 0)
    bm_case("bm_block_method_c1.npz", 112, 112, 32, 0, 8, False, 12, 6, 0)

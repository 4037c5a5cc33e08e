"""The tuned CPU path that bench.py times beside the GPU (oracle/pc_fast.c, `cpu_baseline.tuned`) computes the same
estimator as the f32 oracle: within 1e-4 px on every patch whose arg-max is stable, same validity pattern."""
import os

import numpy as np
import pytest

import oracle_lib as O
from mrs_optic_flow_amd import synth

TOL = 1e-4
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("n,shape,grid,origin,stride", [
    (64, (480, 752), (8, 8), (1, 1), (98, 59)),
    (128, (270, 480), (3, 2), (0, 0), (119, 63)),
    (32, (70, 130), (3, 1), (2, 3), (33, 1)),
])
def test_tuned_path_agrees_with_the_oracle(n, shape, grid, origin, stride):
    h, w = shape
    cur, prev, shifts, kinds = synth.batch_np(8, h, w, n // 8, k0=0)
    lay = O.fft_layout(w, h, n, grid[0], grid[1], origin, stride)
    checked = 0
    for k in range(8):
        want64, _, diags = O.fft_process(cur[k], prev[k], lay, 64, want_diag=True)
        want32, _ = O.fft_process(cur[k], prev[k], lay, 32)
        got = O.fft_process_fast(cur[k], prev[k], lay)
        for p in range(want32.shape[0]):
            stable = diags[p].second_value < 0.5 * diags[p].peak_value or np.allclose(want64[p], want32[p], rtol=0, atol=TOL,
                                                                                      equal_nan=True)
            if stable:
                assert np.allclose(got[p], want32[p], rtol=0, atol=TOL, equal_nan=True), (kinds[k], k, p, got[p], want32[p])
                checked += 1
    assert checked > 0.8 * 8 * grid[0] * grid[1]


def test_tuned_path_on_the_golden_vectors_and_bad_arguments():
    g = np.load(os.path.join(GOLDEN, "fft_n64_unaligned.npz"))
    w, h, n, gx, gy, ox, oy, sx, sy = (int(v) for v in g["layout"])
    lay = O.fft_layout(w, h, n, gx, gy, (ox, oy), (sx, sy))
    ok = g["well_conditioned"]
    for k in range(g["cur"].shape[0]):
        got = O.fft_process_fast(g["cur"][k], g["prev"][k], lay)
        assert np.allclose(got[ok[k]], g["expected"][k][ok[k]], rtol=0, atol=TOL, equal_nan=True)
    with pytest.raises(ValueError):  # 120 is not a power of two: the tuned path declines, it never approximates
        O.fft_process_fast(np.zeros((480, 480), np.uint8), np.zeros((480, 480), np.uint8), O.fft_layout(480, 480, 120, 4, 4))

"""GPU tests of the FFT path at BASELINE's full c4 size (size-independent properties), of the bindings' argument checks and of the
long-range gate."""
import os

import numpy as np
import pytest
import torch

import oracle_lib as O
from mrs_optic_flow_amd import FastSpacedBMMethod, FftMethod, ScaleRotationEstimator, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 1e-4


def _compare(got, cur, prev, lay, label=""):
    """Same rule as tests/test_gpu_fft.py::_compare; returns (checked, total)."""
    want64, _, diags = O.fft_process(cur, prev, lay, 64, want_diag=True)
    want32, _ = O.fft_process(cur, prev, lay, 32)
    n_checked = 0
    for p in range(want64.shape[0]):
        if diags[p].second_value < 0.5 * diags[p].peak_value:
            assert np.allclose(got[p], want64[p], rtol=0, atol=TOL, equal_nan=True), (label, p, got[p], want64[p])
            assert np.allclose(got[p], want32[p], rtol=0, atol=TOL, equal_nan=True), (label, p, got[p], want32[p])
            n_checked += 1
        elif np.array_equal(np.isnan(want64[p]), np.isnan(want32[p])) and np.allclose(want64[p], want32[p], rtol=0,
                                                                                      atol=TOL, equal_nan=True):
            assert np.allclose(got[p], want64[p], rtol=0, atol=TOL, equal_nan=True), (label, "ill", p, got[p], want64[p])
    return n_checked, want64.shape[0]


def test_full_size_c4_batch_properties(gpu):
    """BASELINE config c4 at full size on one GPU's shard: 1920x1080, 16x16 grid of 128x128 patches (persistent
    workgroups, one per CU), batch 64. Planted shift recovered, identical -> 0, pair-alone bit-equality, three pairs
    against the oracle patch by patch."""
    B, h, w, n = 64, 1080, 1920, 128
    cur, prev, shifts, kinds = synth.batch_torch(B, h, w, n // 8, gpu)
    fm = FftMethod(sample_point_size=n, frame_shape=(h, w), grid=(16, 16), origin=(0, 0), stride=(119, 63))
    assert fm.n_patches == 256
    out = fm.process_batch_device(cur, prev)
    torch.cuda.synchronize()
    res = out.cpu().numpy()
    sh = shifts.numpy()
    seen = set()
    for k in range(B):
        seen.add(kinds[k])
        if kinds[k] in ("shift", "noisy"):
            assert np.isfinite(res[k]).all()
            assert np.abs(np.median(res[k], axis=0) - sh[k]).max() < 0.3, (k, kinds[k])
            assert np.abs(res[k] - sh[k]).max() < 1.0
        elif kinds[k] == "identical":
            assert np.abs(res[k]).max() < 1e-4
        else:  # constant frames: (1 - N/2, 1 - N/2) = (-63, -63), |.| > 80 px -> gated to NaN (FftMethod.cpp:1841)
            assert np.isnan(res[k]).all()
    assert {"shift", "identical", "constant", "noisy"} <= seen
    for k in (0, 31, 63):
        alone = fm.process_batch_device(cur[k:k + 1], prev[k:k + 1]).cpu().numpy()[0]
        assert np.array_equal(alone, res[k], equal_nan=True)
    lay = O.fft_layout(w, h, n, 16, 16, (0, 0), (119, 63))
    checked = total = 0
    for k in (1, 30, 61):
        c, t = _compare(res[k], cur[k].cpu().numpy(), prev[k].cpu().numpy(), lay, f"c4/{k}/{kinds[k]}")
        checked, total = checked + c, total + t
    print(f"c4 full size: {checked}/{total} patches well-conditioned and within {TOL} px")
    assert checked > 0.6 * total


def test_bindings_reject_mismatched_frames(gpu):
    """The C ABI receives raw pointers and a pitch, so shapes are checked in the binding: a smaller tensor would make
    the kernels read past the allocation (round-1 advisor finding)."""
    h, w = 96, 160
    small = torch.zeros((2, h - 8, w), dtype=torch.uint8, device=gpu)
    good = torch.zeros((2, h, w), dtype=torch.uint8, device=gpu)
    fm = FftMethod(sample_point_size=64, frame_shape=(h, w), grid=(2, 1), origin=(0, 0), stride=(90, 1))
    bm = FastSpacedBMMethod(16, 8, 8, (h, w))
    sr = ScaleRotationEstimator(240, 40.0)
    lr = FftMethod(512, 64, 80.0)
    for call in (lambda: fm.process_batch_device(small, small),
                 lambda: fm.process_batch_device(good, small),
                 lambda: fm.process_batch_device(good.cpu(), good.cpu()),
                 lambda: fm.process_batch_device(good[0], good[0]),
                 lambda: fm.process_batch_device(good.to(torch.int8), good.to(torch.int8)),
                 lambda: fm.process_batch_device(good[:, :, ::2], good[:, :, ::2]),
                 lambda: fm.process_batch_device(good, good, out=torch.empty(3, device=gpu, dtype=torch.float64)),
                 lambda: bm.process_batch_device(small, small),
                 lambda: bm.process_batch_device(good.cpu(), good.cpu()),
                 lambda: bm.process_batch_host(small.cpu().numpy(), small.cpu().numpy()),
                 lambda: bm.setImPrev(np.zeros((h - 1, w), np.uint8)),
                 lambda: bm.processBlocks(np.zeros((h, w + 1), np.uint8)),
                 lambda: fm.setImPrev(np.zeros((h, w - 1), np.uint8)),
                 lambda: fm.process_batch_host(np.zeros((1, h, w - 2), np.uint8), np.zeros((1, h, w - 2), np.uint8)),
                 lambda: lr.process_long_range_batch_device(good, good),
                 lambda: lr.process_long_range_batch_device(torch.zeros((1, 512, 512), dtype=torch.uint8),
                                                            torch.zeros((1, 512, 512), dtype=torch.uint8)),
                 lambda: sr.process_batch_device(good, good),
                 lambda: sr.logpolar_batch_device(good),
                 lambda: sr.logpolar_batch_device(torch.zeros((1, 240, 240), dtype=torch.uint8, device=gpu), 3)):
        with pytest.raises((ValueError, RuntimeError)) as exc:
            call()
        assert not isinstance(exc.value, AssertionError)
    # and the well-formed calls still work
    assert fm.process_batch_device(good, good).shape == (2, 2, 2)
    assert bm.process_batch_device(good, good)[0].shape[0] == 2


def test_long_range_gate_is_held_in_ints(gpu):
    """`int max_px_speed_lr, max_px_speed_sq_lr` (include/FftMethod.h:393; src/FftMethod.cpp:1687-1688): with
    max_px_speed = 2.9 the long-range gate is (int)2.9 squared = 4, the ordinary gate 8.41. Exact circular shifts of
    the quarter-resolution frame: (1, 1) -> 2 < 4, valid under both; (2, 1) -> 5 > 4, invalid in long-range mode only
    (5 < 8.41). Shifts sitting exactly on a gate (e.g. (2, 0)) are avoided: there rounding noise of 1e-8 px decides."""
    fs, n, speed = 256, 64, 2.9   # sqNum = 4 -> sqNum_lr = 1; the quarter frame is one 64 x 64 patch
    rng = np.random.default_rng(12)
    q_prev = rng.integers(0, 256, (n, n), dtype=np.uint8)
    lay = O.fft_layout(fs, fs, n, 4, 4, max_px_speed=speed)
    for (sx, sy), lr_valid in (((1, 1), True), ((2, 1), False), ((-1, 1), True), ((-1, -2), False)):
        q_cur = np.roll(q_prev, (sy, sx), axis=(0, 1))
        # full-resolution frames whose exact quarter reduction is (q_cur, q_prev): every 4x4 cell constant
        cur = np.kron(q_cur, np.ones((4, 4), np.uint8))
        prev = np.kron(q_prev, np.ones((4, 4), np.uint8))
        assert np.array_equal(O.resize_quarter(cur), q_cur)
        fm = FftMethod(fs, n, speed)
        fm.processImageLongRange(prev)
        got = fm.processImageLongRange(cur)
        want, _ = O.fft_process_long_range(cur, prev, lay, 64)
        assert np.allclose(got, want, rtol=0, atol=TOL, equal_nan=True)
        assert bool(np.isfinite(got).all()) == lr_valid, ((sx, sy), got)
        if lr_valid:
            assert np.allclose(got, [[sx, sy]], rtol=0, atol=3e-5)
        # the ordinary path keeps the double gate pow(max_px_speed_t, 2) (:1686): all four shifts are valid there
        ordinary = FftMethod(n, n, speed).process_batch_host(q_cur[None], q_prev[None])[0]
        assert np.allclose(ordinary, [[sx, sy]], rtol=0, atol=3e-5)
        tq = torch.from_numpy(np.stack([cur, prev])).to(gpu)
        batch = fm.process_long_range_batch_device(tq[:1], tq[1:]).cpu().numpy()[0]
        assert np.array_equal(batch, got, equal_nan=True)

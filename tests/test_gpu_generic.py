"""GPU parity tests of the size-generic paths (round 4): ANY samplePointSize / resolution the reference accepts.

FftMethod takes frameSize / samplePointSize from ROS parameters (/root/reference/src/FftMethod.cpp:1680-1720,
config/default.yaml:31-32) and hands each patch to cv::phaseCorrelate, which zero-pads to getOptimalDFTSize(N) -- possibly an
ODD size (74 -> 75). Sizes without a hand-tuned kernel run the planned kernel (csrc/pc_kernel_generic.hip). Same bars as
tests/test_gpu_fft.py: 1e-4 px against both oracle precisions on well-conditioned patches.
"""
import os

import numpy as np
import pytest
import torch

import oracle_lib as O
from mrs_optic_flow_amd import FftMethod, MofError, synth
from mrs_optic_flow_amd.engine import PEAK_OCL
import tolerances
from test_gpu_fft import TOL

pytestmark = pytest.mark.gpu


def _compare(got, cur, prev, lay, label=""):
    """Every clear-peak patch (second-highest surface value outside the 5 x 5 window < half the peak) against the bars of
    tests/tolerances.py: 1e-4 px against both oracle precisions; a patch that misses that is classified from its input pixels
    (tests/conditioning.py) and held to 1e-4 + 2 x the measured scatter of independent f32 transforms on it, never above 1e-3 px, each one
    recorded. Returns the number of patches pinned."""
    return tolerances.check_frame(got, cur, prev, lay, label)


def _planned_variant(n):
    """r05: transform sizes 60, 96, 100 run the half-tile kernel by default (it beats the full-tile planned kernel there by 10 - 25 %,
    profiles/r05_half_vs_planned_bench_ab.txt); MOF_FFT_HALF=0 keeps the planned kernel (a child process below re-runs them that way)."""
    import os
    half = O.optimal_dft_size(n) in (60, 72, 90, 96, 100, 120) and os.environ.get("MOF_FFT_HALF", "") != "0"
    return "planned-half" if half else "planned"


# even 5-smooth sizes (no padding), sizes that pad to an even size, sizes that pad to an ODD size, odd sizes, small sizes
SIZES = [40, 48, 60, 80, 96, 100, 16, 20, 24, 36, 72, 90, 108, 125, 135,  # M = N
         62, 98, 118, 34, 66, 130,                                       # N -> even M (64, 100, 120, 36, 72, 135 is odd)
         74, 44, 26, 124, 134,                                           # N -> odd M (75, 45, 27, 125, 135)
         15, 25, 27, 45, 75, 33, 51]                                     # odd N


@pytest.mark.parametrize("n", SIZES)
def test_planned_kernel_matches_oracle_at_every_size(gpu, n):
    gx, gy = (3, 2) if n <= 100 else (2, 2)
    stride = (n + 3, n + 1)
    w, h = 5 + stride[0] * (gx - 1) + n + 2, 3 + stride[1] * (gy - 1) + n + 1
    B = 6
    cur, prev, shifts, kinds = synth.batch_np(B, h, w, max(1, n // 8), k0=n)
    fm = FftMethod(sample_point_size=n, frame_shape=(h, w), grid=(gx, gy), origin=(5, 3), stride=stride)
    assert fm.kernel_variant == _planned_variant(n)
    got = fm.process_batch_device(torch.from_numpy(cur).to(gpu), torch.from_numpy(prev).to(gpu)).cpu().numpy()
    lay = O.fft_layout(w, h, n, gx, gy, (5, 3), stride)
    checked = sum(_compare(got[k], cur[k], prev[k], lay, f"n{n}/pair{k}/{kinds[k]}") for k in range(B))
    assert checked > (0.6 if n >= 32 else 0.15) * B * gx * gy, (n, checked)  # (tiny patches: few surfaces have a runner-up below half the peak)
    for k in range(B):
        if kinds[k] == "shift" and n >= 40:  # (small patches under a non-circular shift are biased towards zero: the oracle comparison above is the test)
            # odd transform sizes carry cv::phaseCorrelate's half-pixel centre (M / 2.0 against the integer M >> 1)
            bias = 0.5 if O.optimal_dft_size(n) % 2 else 0.0
            assert np.allclose(np.nanmedian(got[k], axis=0) + bias, shifts[k], rtol=0, atol=0.5), (n, k)


@pytest.mark.parametrize("fs,n", [(480, 60), (480, 80), (480, 96), (400, 100), (480, 40), (480, 48), (296, 74), (480, 30)])
def test_reference_tiling_stateful_entry(gpu, fs, n):
    """new FftMethod(frame_size, sample_point_size, ...) as the node constructs it (optic_flow.cpp:1001-1002), driven through
    processImage: first frame against itself, then against its predecessor."""
    fm = FftMethod(fs, n, 80.0)
    sq = fs // n
    assert fm.sqNum == sq
    # (1 6 1)-blurred texture under a seed with no exactly-zero spectral bin on any patch: tests/test_conditioning.py::tiling_frames)
    from test_conditioning import tiling_frames
    seq = tiling_frames(fs, n)
    lay = O.fft_layout(fs, fs, n, sq, sq)
    out0 = fm.processImage(seq[0])
    assert np.allclose(out0, O.fft_process(seq[0], seq[0], lay, 64)[0], rtol=0, atol=TOL, equal_nan=True)
    for t in (1, 2):
        out = fm.processImage(seq[t])
        assert _compare(out, seq[t], seq[t - 1], lay, f"fs{fs}/n{n}/t{t}") > 0.7 * sq * sq


@pytest.mark.parametrize("n", [60, 96, 100, 74])
def test_planned_kernel_front_ends(gpu, n):
    """The three front ends of K1 on a planned size: BGR8 frames (CV_RGB2GRAY fused, optic_flow.cpp:1622), the long-range mode
    (quarter-resolution pixels formed on the fly, FftMethod.cpp:1905-2007) and a video through the sequence entry."""
    rng = np.random.default_rng(n)
    # BGR
    gx, gy = 2, 2
    w, h = 2 * n + 9, 2 * n + 5
    B = 3
    bgr_c = rng.integers(0, 256, (B, h, w, 3), dtype=np.uint8)
    bgr_p = np.roll(bgr_c, (2, -3), axis=(1, 2))
    fm = FftMethod(sample_point_size=n, frame_shape=(h, w), grid=(gx, gy), origin=(1, 2), stride=(n + 4, n + 2))
    got = fm.process_batch_device_bgr(torch.from_numpy(bgr_c).to(gpu), torch.from_numpy(bgr_p).to(gpu)).cpu().numpy()
    gray_c = np.stack([O.rgb2gray(f) for f in bgr_c])
    gray_p = np.stack([O.rgb2gray(f) for f in bgr_p])
    same = fm.process_batch_device(torch.from_numpy(gray_c).to(gpu), torch.from_numpy(gray_p).to(gpu)).cpu().numpy()
    assert np.array_equal(got, same, equal_nan=True)
    lay = O.fft_layout(w, h, n, gx, gy, (1, 2), (n + 4, n + 2))
    assert sum(_compare(got[k], gray_c[k], gray_p[k], lay, f"bgr{k}") for k in range(B)) >= B * gx * gy - 2
    # long range: reference tiling with sqNum = 4 -> one quarter-resolution patch
    fs = 4 * n
    flr = FftMethod(fs, n, 80.0)
    cur, prev = synth.pair_np(n, fs, fs, 8, -12, blur=True)
    out = flr.process_long_range_batch_device(torch.from_numpy(cur[None]).to(gpu), torch.from_numpy(prev[None]).to(gpu)).cpu().numpy()[0]
    want, _ = O.fft_process_long_range(cur, prev, O.fft_layout(fs, fs, n, 4, 4), 64)
    assert np.allclose(out, want, rtol=0, atol=TOL, equal_nan=True), (out, want)
    # a video: pair k = (frame k + 1, frame k)
    video = np.stack([synth.pair_np(3 * n, h, w, 2 * t, t, blur=True)[0] for t in range(4)])
    seq = fm.process_sequence_device(torch.from_numpy(video).to(gpu)).cpu().numpy()
    for k in range(3):
        assert _compare(seq[k], video[k + 1], video[k], lay, f"seq{k}") >= gx * gy - 1


def test_planned_kernel_degenerate_pairs_and_gate(gpu):
    """Constant patches (flat surface: the first-index / clamped-centroid artefact), black frames on a PADDED size (the only
    constant patch that stays constant after cv::phaseCorrelate's zero padding) and the gate against samplePointSize / 2."""
    for n in (60, 96):  # M = N: the closed form (the oracle's own radix-3/5 DFT of a constant is not exactly zero off DC)
        tex = synth.canvas_np(5, n, n, True)[:n, :n].copy()
        fm = FftMethod(n, n, 80.0)
        for c in (200, 0):
            const = np.full((n, n), c, np.uint8)
            for cur, prev in ((const, const), (const, tex), (tex, const)):
                out = fm.process_batch_host(cur[None], prev[None])[0]
                P = float(cur.astype(np.float64).sum()) * float(prev.astype(np.float64).sum())
                c9 = 9.0 * (P / (P * P + float(np.finfo(np.float32).eps))) if P > 0 else 0.0
                want = (c9 / (c9 + float(np.finfo(np.float64).eps)) if c9 > 0 else 0.0) - n / 2
                assert np.allclose(out, [[want, want]], rtol=0, atol=1e-4), (n, c, out, want)
    for n in (62, 74):  # padded: an all-zero patch gives shift -M/2, beyond N/2 -> invalid; oracle agrees
        tex = synth.canvas_np(6, n, n, True)[:n, :n].copy()
        zero = np.zeros((n, n), np.uint8)
        fm = FftMethod(n, n, 80.0)
        lay = O.fft_layout(n, n, n, 1, 1)
        for cur, prev in ((zero, zero), (zero, tex), (tex, zero)):
            out = fm.process_batch_host(cur[None], prev[None])[0]
            want, _ = O.fft_process(cur, prev, lay, 64)
            assert np.isnan(want).all() and np.isnan(out).all(), (n, out, want)
    # the gate: a 7-px shift passes max_px_speed 7.5 and fails 6.5; |shift| <= N/2 is judged on the unpadded size
    n = 60
    prev = synth.canvas_np(3, n, n, False)[:n, :n].copy()
    cur = np.roll(prev, (0, 7), axis=(0, 1))
    assert np.isnan(FftMethod(n, n, 6.5).process_batch_host(cur[None], prev[None])).all()
    assert np.allclose(FftMethod(n, n, 7.5).process_batch_host(cur[None], prev[None])[0], [[7.0, 0.0]], rtol=0, atol=1e-5)


def test_ocl_peak_model_on_planned_sizes(gpu):
    """MOF_PEAK_OCL on 5-smooth even sizes without a tuned kernel; sizes the reference's OpenCL branch cannot plan are refused."""
    n, gx, gy = 60, 2, 2
    w, h = 2 * n + 6, 2 * n + 4
    cur, prev, _, kinds = synth.batch_np(4, h, w, 5, k0=3)
    fm = FftMethod(sample_point_size=n, frame_shape=(h, w), grid=(gx, gy), origin=(2, 1), stride=(n + 2, n + 1), peak_model=PEAK_OCL,
                   search_radius=55)
    got = fm.process_batch_device(torch.from_numpy(cur).to(gpu), torch.from_numpy(prev).to(gpu)).cpu().numpy()
    lay = O.fft_layout(w, h, n, gx, gy, (2, 1), (n + 2, n + 1))
    n_ok = 0
    for k in range(4):
        want64, _, diags = O.fft_process_ocl(cur[k], prev[k], lay, 55, 64, want_diag=True)
        for p in range(gx * gy):
            if diags[p].second_value < 0.5 * diags[p].peak_value:
                assert np.allclose(got[k][p], want64[p], rtol=0, atol=TOL, equal_nan=True), (k, p, got[k][p], want64[p])
                n_ok += 1
    assert n_ok >= 8
    for bad in (62, 74, 45):
        with pytest.raises(MofError) as exc:
            FftMethod(bad, bad, 80.0, peak_model=PEAK_OCL)
        assert exc.value.code == -5


# ---- patches too large for one CU's full tile (padded side > 135): the fused half-tile kernel (csrc/pc_half_kernel.hip, r05) where
#      the HALF tile fits -- even padded sizes up to 192 --, the planned pipeline through HBM scratch (csrc/pc_large_kernel.hip) beyond ----
def _large_variant(n):
    import os
    m = O.optimal_dft_size(n)
    half = m in (144, 150, 160, 162, 180, 192) and os.environ.get("MOF_FFT_HALF", "") != "0"
    return "planned-half" if half else "planned-large"


@pytest.mark.parametrize("n", [160, 136, 144, 150, 180, 192, 162, 158, 170, 186, 200, 216, 240, 250, 256, 148, 225, 243, 202, 196, 230, 252, 288, 320, 360, 384, 280, 310, 340, 380, 300, 270, 450, 262, 296, 440, 400, 432, 246, 390, 420, 480, 512, 505, 750, 810, 324, 486, 500, 540, 576, 600, 640, 648, 720, 768, 800, 864, 900, 960, 530, 700, 930, 490, 375, 405, 625, 675, 729, 220, 370, 610])  # (750, 810: one stage body per radix;
                                                                                                               #  240 / 256 / 480: the estimator's tuned transforms)
def test_large_patches_match_oracle(gpu, n):
    gx, gy = (2, 2) if n <= 256 else (1, 1)
    stride = (n + 5, n + 2)
    w, h = 3 + stride[0] * (gx - 1) + n + 4, 2 + stride[1] * (gy - 1) + n + 3
    B = 5 if n <= 256 else 3
    cur, prev, shifts, kinds = synth.batch_np(B, h, w, min(n // 8, 24), k0=n)
    fm = FftMethod(sample_point_size=n, frame_shape=(h, w), grid=(gx, gy), origin=(3, 2), stride=stride)
    assert fm.kernel_variant == _large_variant(n)
    got = fm.process_batch_device(torch.from_numpy(cur).to(gpu), torch.from_numpy(prev).to(gpu)).cpu().numpy()
    lay = O.fft_layout(w, h, n, gx, gy, (3, 2), stride)
    checked = sum(_compare(got[k], cur[k], prev[k], lay, f"n{n}/pair{k}/{kinds[k]}") for k in range(B))
    assert checked > 0.6 * B * gx * gy, (n, checked)
    for k in range(B):
        if kinds[k] == "shift":
            bias = 0.5 if O.optimal_dft_size(n) % 2 else 0.0
            assert np.allclose(np.nanmedian(got[k], axis=0) + bias, shifts[k], rtol=0, atol=0.5), (n, k)
    # one pair at a time (a pass of one frame pair through the scratch) gives the same bits
    one = fm.process_batch_device(torch.from_numpy(cur[1:2]).to(gpu), torch.from_numpy(prev[1:2]).to(gpu)).cpu().numpy()
    assert np.array_equal(one[0], got[1], equal_nan=True)


@pytest.mark.parametrize("fs,n", [(480, 160), (480, 240), (480, 480), (450, 150), (470, 100), (400, 200)])
def test_large_patches_reference_constructor_and_stateful_entry(gpu, fs, n):
    """FftMethod(frame_size, sample_point_size) as the node constructs it, incl. the reference's fallback to ONE patch = the whole
    frame when frameSize is not a multiple of samplePointSize (FftMethod.cpp:1709-1716: 470 / 100 -> one 470 x 470 patch)."""
    fm = FftMethod(fs, n, 80.0)
    sps = n if fs % n == 0 else fs
    sq = fs // sps
    assert fm.cfg.patch_size == sps and fm.sqNum == sq and fm.kernel_variant == _large_variant(sps)
    seq = [synth.pair_np(7 + n, fs, fs, 3 * t, -2 * t, blur=True)[0] for t in range(3)]
    lay = O.fft_layout(fs, fs, sps, sq, sq)
    out0 = fm.processImage(seq[0])
    assert np.allclose(out0, O.fft_process(seq[0], seq[0], lay, 64)[0], rtol=0, atol=TOL, equal_nan=True)
    for t in (1, 2):
        out = fm.processImage(seq[t])
        assert _compare(out, seq[t], seq[t - 1], lay, f"fs{fs}/n{n}/t{t}") >= max(1, sq * sq - 1)
    # degenerate pairs: a black frame against texture (flat zero surface -> shift -M/2, valid only if it passes the gate)
    black = np.zeros((fs, fs), np.uint8)
    for cur, prev in ((black, seq[0]), (seq[0], black), (black, black)):
        got = fm.process_batch_host(cur[None], prev[None])[0]
        want, _ = O.fft_process(cur, prev, lay, 64)
        assert np.allclose(got, want, rtol=0, atol=TOL, equal_nan=True), (got, want)


def test_large_patches_front_ends_and_passes(gpu):
    """BGR8 frames, the long-range mode and a video on a large patch size; a batch that spans several passes of the scratch
    (MOF_FFT_LARGE_PASS is not set: the pass size follows from the 1.5 GB budget, so force small passes through a second engine
    that has only seen small batches)."""
    n, gx, gy = 160, 2, 1
    w, h = 2 * n + 7, n + 5
    rng = np.random.default_rng(5)
    B = 4
    bgr_c = rng.integers(0, 256, (B, h, w, 3), dtype=np.uint8)
    bgr_p = np.roll(bgr_c, (3, -5), axis=(1, 2))
    fm = FftMethod(sample_point_size=n, frame_shape=(h, w), grid=(gx, gy), origin=(2, 3), stride=(n + 3, 1))
    got = fm.process_batch_device_bgr(torch.from_numpy(bgr_c).to(gpu), torch.from_numpy(bgr_p).to(gpu)).cpu().numpy()
    gray_c = np.stack([O.rgb2gray(f) for f in bgr_c])
    gray_p = np.stack([O.rgb2gray(f) for f in bgr_p])
    same = fm.process_batch_device(torch.from_numpy(gray_c).to(gpu), torch.from_numpy(gray_p).to(gpu)).cpu().numpy()
    assert np.array_equal(got, same, equal_nan=True)
    lay = O.fft_layout(w, h, n, gx, gy, (2, 3), (n + 3, 1))
    assert sum(_compare(got[k], gray_c[k], gray_p[k], lay, f"bgr{k}") for k in range(B)) >= B * gx * gy - 1
    video = np.stack([synth.pair_np(77, h, w, 3 * t, t, blur=True)[0] for t in range(4)])
    seq = fm.process_sequence_device(torch.from_numpy(video).to(gpu)).cpu().numpy()
    for k in range(3):
        assert _compare(seq[k], video[k + 1], video[k], lay, f"seq{k}") >= gx * gy - 1
    fs = 4 * n
    flr = FftMethod(fs, n, 80.0)
    cur, prev = synth.pair_np(n, fs, fs, 8, -12, blur=True)
    out = flr.process_long_range_batch_device(torch.from_numpy(cur[None]).to(gpu), torch.from_numpy(prev[None]).to(gpu)).cpu().numpy()[0]
    want, _ = O.fft_process_long_range(cur, prev, O.fft_layout(fs, fs, n, 4, 4), 64)
    assert np.allclose(out, want, rtol=0, atol=TOL, equal_nan=True), (out, want)


def test_patches_of_200_pixels_on_the_tuned_transforms(gpu):
    """r06: unpadded 200 x 200 patches run the estimator's tuned transforms (csrc/sr_common.hpp: SrPlan<200> = 10 x 20;
    25 one-wave row workgroups per image, a thirteenth candidate workgroup for the 100 row pairs) -- gray and BGR8 frames give the same
    bits, a batch and its pairs one at a time too, every clear-peak patch within the bars of both oracles; 198-pixel patches pad to 200
    and keep the planned pipeline."""
    n, gx, gy = 200, 2, 1
    w, h = 2 * n + 9, n + 6
    rng = np.random.default_rng(200)
    B = 3
    bgr_c = rng.integers(0, 256, (B, h, w, 3), dtype=np.uint8)
    bgr_p = np.roll(bgr_c, (4, -7), axis=(1, 2))
    fm = FftMethod(sample_point_size=n, frame_shape=(h, w), grid=(gx, gy), origin=(3, 2), stride=(n + 4, 1))
    assert fm.kernel_variant == "planned-large"
    got = fm.process_batch_device_bgr(torch.from_numpy(bgr_c).to(gpu), torch.from_numpy(bgr_p).to(gpu)).cpu().numpy()
    gray_c = np.stack([O.rgb2gray(f) for f in bgr_c])
    gray_p = np.stack([O.rgb2gray(f) for f in bgr_p])
    same = fm.process_batch_device(torch.from_numpy(gray_c).to(gpu), torch.from_numpy(gray_p).to(gpu)).cpu().numpy()
    assert np.array_equal(got, same, equal_nan=True)
    lay = O.fft_layout(w, h, n, gx, gy, (3, 2), (n + 4, 1))
    assert sum(_compare(got[k], gray_c[k], gray_p[k], lay, f"t200/bgr{k}") for k in range(B)) >= B * gx * gy - 1
    for k in range(B):
        one = fm.process_batch_device(torch.from_numpy(gray_c[k:k + 1]).to(gpu), torch.from_numpy(gray_p[k:k + 1]).to(gpu)).cpu().numpy()
        assert np.array_equal(one[0], same[k], equal_nan=True)
    # circular shifts of a 200 x 200 texture: exact integers while the 5 x 5 window lies inside the surface; at +-99 the window is clamped
    # at the border (weightedCentroid, FftMethod.cpp:1337-1383) and the answer is the oracle's -- the peak then sits in the LAST candidate
    # workgroup of K7 (100 row pairs = 12 workgroups of 8 and one of 4)
    tex = synth.canvas_np(9, n, n, False)[:n, :n].copy()
    f1 = FftMethod(n, n, 200.0)
    lay1 = O.fft_layout(n, n, n, 1, 1, (0, 0), (n, n), 200.0)
    for dx, dy in ((0, 0), (37, -61), (-99, 99), (5, 99), (-98, -99)):
        cur = np.roll(tex, (dy, dx), axis=(0, 1))
        out = f1.process_batch_host(cur[None], tex[None])[0]
        want, _ = O.fft_process(cur, tex, lay1, 64)
        assert np.allclose(out, want, rtol=0, atol=1e-4, equal_nan=True), (dx, dy, out, want)
        if max(abs(dx), abs(dy)) <= 97:
            assert np.allclose(out, [[dx, dy]], rtol=0, atol=1e-4), (dx, dy, out)
    video = np.stack([synth.pair_np(21, h, w, 2 * t, -t, blur=True)[0] for t in range(3)])
    seq = fm.process_sequence_device(torch.from_numpy(video).to(gpu)).cpu().numpy()
    for k in range(2):
        assert _compare(seq[k], video[k + 1], video[k], lay, f"t200/seq{k}") >= gx * gy - 1
    # the input classes whose alternating-sign pixel sums cancel exactly (synth.fuzz_classes_np "checker": the real-only CCS slots must come
    # out of the tuned two-stage transforms exactly -- bin N/2 of butterfly20 / butterfly18 passes no twiddle), saturated and smooth content
    for m in (200, 216, 250, 270, 400, 432):  # (250, 400, 432: odd last radix -- their real-only slots come from the images' exact integer sums)
        fc = FftMethod(m, m, 80.0)
        cls = synth.fuzz_classes_np(300 + m, m, m, 3, -2)
        lay_m = O.fft_layout(m, m, m, 1, 1)
        for name in ("checker", "saturated", "const_rect", "smooth"):
            c, p = cls[name]
            out = fc.process_batch_host(c[None], p[None])[0]
            tolerances.check_frame(out, c, p, lay_m, f"t{m}/{name}")
        c, p = cls["checker"]
        want, _ = O.fft_process(c, p, lay_m, 64)
        assert np.allclose(fc.process_batch_host(c[None], p[None])[0], want, rtol=0, atol=1e-4, equal_nan=True)


_LARGE_VIDEO_SCRIPT = r"""
import sys
sys.path[:0] = [{root!r}, {tests!r}]
import numpy as np, torch
from mrs_optic_flow_amd import FftMethod, synth
dev = torch.device("cuda", 0)
checked = 0
for n, gx in ((200, 2), (196, 2), (252, 1), (216, 1), (310, 1), (300, 1), (266, 1), (250, 1), (394, 1), (540, 1), (715, 1), (222, 1), (405, 1)):
    w, h = gx * (n + 4) + 5, n + 6
    F = 11
    video, _ = synth.video_torch(F, h, w, "cpu", k=n)
    video[4] = 93          # a constant frame in the stream (padded sizes: the box-zero rule through the video form's flags)
    video[8] = video[7]    # a repeated frame
    for ch in (1, 3):
        fm = FftMethod(sample_point_size=n, frame_shape=(h, w), grid=(gx, 1), origin=(2, 3), stride=(n + 4, 1))
        if ch == 1:
            dv = video.to(dev)
            seq = fm.process_sequence_device(dv)
            pair = fm.process_batch_device(dv[1:].clone(), dv[:-1].clone())   # (separate allocations: the pair form)
        else:
            dv = video.to(dev)[..., None].expand(-1, -1, -1, 3).contiguous()
            seq = fm.process_sequence_device_bgr(dv)
            pair = fm.process_batch_device_bgr(dv[1:].clone(), dv[:-1].clone())
        assert torch.equal(torch.nan_to_num(seq, nan=-1e9), torch.nan_to_num(pair, nan=-1e9)), (n, ch)
        assert torch.equal(torch.isnan(seq), torch.isnan(pair))
        checked += 1
print("large video ok", checked)
"""


def test_video_form_of_the_tuned_large_patch_path(gpu):
    """r06: a video on patches of 193 .. 256 / 451 .. 480 pixels forms every frame's row spectra ONCE per pass (Zh slot = frame x patches +
    patch; the column kernel finds pair q's images at slots q and q + patches; the per-image flags are re-laid per pair for the box-zero
    rule and the tail). Same kernels, same arithmetic: the video entry must return the pair entry's BITS -- gray and BGR8, unpadded and
    padded sizes, a constant and a repeated frame in the stream, passes of 3 pairs (MOF_FFT_LARGE_PASS) so that the 10 pairs take four
    passes of the scratch; and MOF_FFT_LARGE_VIDEO=0 keeps the pair form."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = _LARGE_VIDEO_SCRIPT.format(root=root, tests=os.path.join(root, "tests"))
    for env in ({"MOF_FFT_LARGE_PASS": "3"}, {}, {"MOF_FFT_LARGE_VIDEO": "0"}):
        r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
        assert r.returncode == 0 and "large video ok 26" in r.stdout, (env, r.stdout[-1500:], r.stderr[-2500:])


# ---- scaleRotationEstimator at any even resolution (scaleRotationEstimator.cpp:3-32) ----
@pytest.mark.parametrize("res,M", [(320, 45.0), (360, 49.9), (400, 49.9), (128, 25.0), (200, 35.0), (250, 40.0), (300, 49.9),
                                    (350, 49.9), (500, 60.0), (192, 30.0), (640, 70.0), (720, 75.0), (270, 40.0), (208, 35.0), (336, 49.9), (416, 55.0), (432, 55.0), (496, 60.0), (160, 30.0), (180, 30.0), (144, 28.0), (220, 35.0), (370, 50.0), (100, 22.0), (120, 25.0), (150, 28.0), (104, 22.0)])
def test_scale_rotation_at_any_resolution(gpu, res, M):
    """r06: 200, 270, 300, 320, 360, 500, 640, 720 are tuned transform sizes of the FFT engine's large patches with an exact Nyquist bin -- the
    estimator runs K5s / K6s / K7 there; 208 (-> 216), 336 / 350 (-> 360), 416 (-> 432), 496 (-> 500) PAD to a tuned size and 250 / 400 / 432 are the
    odd-last-radix sizes (exact pixel sums inside the Zh slots): K5s' zero-padding frame form / K6s / K7 + the planned final kernel. 128, 160, 180, 192: tuned plans of their own (8 / 10 / 10 / 12 x 16 or 18); 144:
    the planned pipeline (MOF_SR_TUNED_ALL=0 in the child test below: everywhere)."""
    import sr_scenes
    from mrs_optic_flow_amd import ScaleRotationEstimator
    from mrs_optic_flow_amd.engine import INTER_CUBIC, INTER_LANCZOS4

    base = sr_scenes.canvas(11 + res, res)
    params = [(1.0, 0.0), (1.03, 2.0), (0.96, -3.0), (1.0, 5.0), (1.05, -1.0)]
    frames = np.stack([sr_scenes.view(base, res, s, r) for s, r in params])
    est = ScaleRotationEstimator(res, M)
    tv = torch.from_numpy(frames).to(gpu)
    # the remap, byte for byte
    lp = est.logpolar_batch_device(tv, INTER_LANCZOS4).cpu().numpy()
    lc = est.logpolar_batch_device(tv[:2], INTER_CUBIC).cpu().numpy()
    for k in range(len(params)):
        assert np.array_equal(lp[k], O.logpolar(frames[k], M, INTER_LANCZOS4)), (res, k)
    for k in range(2):
        assert np.array_equal(lc[k], O.logpolar(frames[k], M, INTER_CUBIC)), (res, k)
    # independent pairs (each the two-call sequence of a fresh estimator)
    got = est.process_batch_device(tv[1:], tv[:-1]).cpu().numpy()
    for k in range(len(params) - 1):
        ref = O.ScaleRotationEstimator(res, M, 64)
        ref.processImage(frames[k])
        s, r = ref.processImage(frames[k + 1])
        assert np.allclose(got[k, 2:], ref.pt, rtol=0, atol=1e-4), (res, k, got[k], ref.pt)
        assert abs(got[k, 0] - s) < 1e-5 and abs(got[k, 1] - r) < 1e-5
    # the stateful entry and the sequence entry: same bits as each other, the oracle's frame-by-frame loop within tolerance
    one, ref = ScaleRotationEstimator(res, M), O.ScaleRotationEstimator(res, M, 64)
    seq = ScaleRotationEstimator(res, M).process_sequence_device(tv).cpu().numpy()
    for t in range(len(params)):
        s1, r1 = one.processImage(frames[t])
        ws, wr = ref.processImage(frames[t])
        assert (s1, r1) == (seq[t, 0], seq[t, 1]), (res, t)
        assert abs(s1 - ws) < 1e-5 and abs(r1 - wr) < 1e-5, (res, t, s1, r1, ws, wr)


def test_scale_rotation_planned_pipeline_at_the_tuned_sizes(gpu):
    """MOF_SR_TUNED_ALL=0: the estimator keeps the planned pipeline at the resolutions that otherwise take the tuned transforms -- the same
    parity test, in a child process (the knob is read once)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_generic.py"), "-q", "-x", "-m", "gpu", "-k",
                          "scale_rotation_at_any_resolution and (320 or 500 or 270 or 640 or 350 or 400 or 250 or 128 or 192)", "-p", "no:cacheprovider"],
                         env=dict(os.environ, MOF_SR_TUNED_ALL="0"), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and " passed" in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]


def test_planned_kernel_run_time_form(gpu):
    """The run-time-plan form of the planned kernel (what sizes below 16 and the BGR / long-range / OpenCL-model front ends run)
    on sizes that normally take a compile-time instantiation: MOF_PLANNED_STATIC=0 in a child process (the knob is read once)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_generic.py"), "-m", "gpu", "-x", "-q",
                        "-k", "every_size and (60 or 62 or 74 or 96 or 135 or 45)", "-p", "no:cacheprovider"], capture_output=True,
                       text=True, timeout=900, cwd=root, env=dict(os.environ, MOF_PLANNED_STATIC="0", MOF_FFT_HALF="0"))
    assert r.returncode == 0 and " passed" in r.stdout, (r.stdout[-2000:], r.stderr[-1000:])
    # ... and the compile-time-plan form of the full-tile planned kernel on the sizes that run the half-tile kernel by default (r05)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_generic.py"), "-m", "gpu", "-x", "-q",
                        "-k", "(every_size and (60 or 62 or 96 or 98 or 100)) or planned_kernel_front_ends or reference_tiling", "-p", "no:cacheprovider"],
                       capture_output=True, text=True, timeout=900, cwd=root, env=dict(os.environ, MOF_FFT_HALF="0"))
    assert r.returncode == 0 and " passed" in r.stdout, (r.stdout[-2000:], r.stderr[-1000:])


@pytest.mark.parametrize("n", [240, 256])
def test_large_patches_of_the_estimators_sizes_front_ends(gpu, n):
    """Unpadded patches of 240 / 256 / 480 pixels run the estimator's tuned K5s / K6s / K7 under the large-patch pipeline (mof_capi.hip,
    launch_large): BGR8 frames give the gray path's bits there too, a video gives the pair entry's, a constant patch its closed form."""
    gx, gy = 2, 1
    w, h = 2 * n + 9, n + 6
    rng = np.random.default_rng(n)
    B = 3
    bgr_c = rng.integers(0, 256, (B, h, w, 3), dtype=np.uint8)
    bgr_p = np.roll(bgr_c, (2, -3), axis=(1, 2))
    bgr_p[1, 3:3 + n, 1:1 + n] = (40, 90, 200)  # pair 1, patch 0: a constant previous patch
    fm = FftMethod(sample_point_size=n, frame_shape=(h, w), grid=(gx, gy), origin=(1, 3), stride=(n + 5, 1))
    got = fm.process_batch_device_bgr(torch.from_numpy(bgr_c).to(gpu), torch.from_numpy(bgr_p).to(gpu)).cpu().numpy()
    gray_c = np.stack([O.rgb2gray(f) for f in bgr_c])
    gray_p = np.stack([O.rgb2gray(f) for f in bgr_p])
    same = fm.process_batch_device(torch.from_numpy(gray_c).to(gpu), torch.from_numpy(gray_p).to(gpu)).cpu().numpy()
    assert np.array_equal(got, same, equal_nan=True)
    lay = O.fft_layout(w, h, n, gx, gy, (1, 3), (n + 5, 1))
    # (random BGR frames against their rolled copy: patch 0 of pair 1 holds a constant previous patch -- its closed form is pinned in
    #  test_gpu_fft_classes.py; every clear-peak patch goes through the common bars)
    assert sum(_compare(got[k], gray_c[k], gray_p[k], lay, f"tuned-large{n}/bgr{k}") for k in range(B)) >= B * gx * gy - 2
    video = np.stack([synth.pair_np(5 + n, h, w, 2 * t, -t, blur=True)[0] for t in range(4)])
    seq = fm.process_sequence_device(torch.from_numpy(video).to(gpu)).cpu().numpy()
    pairs = fm.process_batch_device(torch.from_numpy(video[1:]).to(gpu), torch.from_numpy(video[:-1]).to(gpu)).cpu().numpy()
    assert np.array_equal(seq, pairs, equal_nan=True)
    for k in range(3):
        assert _compare(seq[k], video[k + 1], video[k], lay, f"tuned-large{n}/seq{k}") >= gx * gy - 1


def test_large_patches_planned_kernels_at_the_estimators_sizes(gpu):
    """MOF_FFT_LARGE_TUNED=0 keeps the planned L5 / L6 / L7 for 200 / 216 / 240 / 256 / 480 (the A/B form): a child process re-runs the large-patch
    parity cases of those sizes with the knob."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, MOF_FFT_LARGE_TUNED="0")
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-k",
                          "large_patches_match_oracle and (200 or 216 or 240 or 256 or 480 or 202 or 196 or 230 or 252 or 288 or 320 or 360 or 384 or 280 or 310 or 340 or 380 or 300 or 270 or 450 or 262 or 296 or 440 or 512 or 505 or 250 or 400 or 432 or 246 or 390 or 420 or 324 or 486 or 500 or 540 or 600 or 640 or 720 or 750 or 810 or 900 or 960 or 530 or 930 or 225 or 243 or 375 or 729 or 220)", "-p", "no:cacheprovider"], env=env, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert " passed" in out.stdout


def test_large_patches_pipeline_where_the_half_tile_kernel_is_the_default(gpu):
    """MOF_FFT_HALF=0 keeps patches of 136 .. 192 pixels on the four-kernel pipeline through HBM scratch (the r04 form, the A/B partner of
    csrc/pc_half_kernel.hip): a child process re-runs their parity cases, the constructor / stateful cases and the front ends that way."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, MOF_FFT_HALF="0")
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-k",
                          "(large_patches_match_oracle and (136 or 144 or 150 or 160 or 162 or 180 or 192)) or large_patches_reference_constructor "
                          "or large_patches_front_ends_and_passes", "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert " passed" in out.stdout


def test_half_tile_formulation_on_the_tuned_sizes(gpu):
    """MOF_FFT_HALF=1 routes 64 / 96 / 120 / 128 through the half-tile kernel too (at 120 / 128 two workgroups share a CU; the A/B of
    VERDICT r04 item 2): a child process runs the oracle comparison of tools/check_half.py on those sizes -- 1e-4 px on every clear-peak
    patch, NaN patterns equal."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "check_half.py"), "64", "96", "120", "128"], env=dict(os.environ, MOF_FFT_HALF="1"),
                         capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0 and "OK 0" in out.stdout and out.stdout.count("variant=planned-half") == 4, out.stdout[-2000:] + out.stderr[-2000:]

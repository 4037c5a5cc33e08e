"""GPU parity tests of the half-tile kernel K1h (csrc/pc_half_kernel.hip): pair entries, the video form (a frame's spectrum kept in
registers), and the planned sizes whose video form it serves -- against both oracles under the bars of tests/tolerances.py."""
import os
import subprocess

import numpy as np
import pytest
import torch

import oracle_lib as O
from mrs_optic_flow_amd import FftMethod, synth

pytestmark = pytest.mark.gpu
TOL = 1e-4
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("n", [160, 150, 137, 186])
def test_half_tile_kernel_entries(gpu, n):
    """csrc/pc_half_kernel.hip is what patches of 136 .. 192 pixels run by default (r05): the batch entry on a strided view (pitch >
    width), one pair alone = the same bits as inside a batch, the video entry = the pair entry's bits. (Stateful entry, BGR8 frames, black
    frames and constant boxes on these sizes: the large-patch tests of test_gpu_generic.py / test_gpu_fft_classes.py, which now reach this kernel.)"""
    gx, gy = 2, 1
    w, h = 2 * n + 11, n + 4
    B = 4
    cur, prev, _, kinds = synth.batch_np(B, h, w + 8, min(n // 8, 20), k0=n + 1)
    fm = FftMethod(sample_point_size=n, frame_shape=(h, w), grid=(gx, gy), origin=(3, 2), stride=(n + 6, 1))
    assert fm.kernel_variant == "planned-half"
    tc, tp = torch.from_numpy(cur).to(gpu)[:, :, :w], torch.from_numpy(prev).to(gpu)[:, :, :w]  # (views: pitch = w + 8)
    got = fm.process_batch_device(tc, tp).cpu().numpy()
    lay = O.fft_layout(w, h, n, gx, gy, (3, 2), (n + 6, 1))
    import tolerances
    for k in range(B):
        c, p = np.ascontiguousarray(cur[k][:, :w]), np.ascontiguousarray(prev[k][:, :w])
        want64, _, diags = O.fft_process(c, p, lay, 64, want_diag=True)
        want32, _ = O.fft_process(c, p, lay, 32)
        for q in range(gx * gy):
            if diags[q].second_value < 0.5 * diags[q].peak_value:
                tolerances.check_patch(got[k][q], want64[q], want32[q], f"half{n}/pair{k}/{kinds[k]}", q, pixels=tolerances.patch_pixels(c, p, lay, q))
    one = fm.process_batch_device(tc[2:3], tp[2:3]).cpu().numpy()
    assert np.array_equal(one[0], got[2], equal_nan=True)
    video = np.stack([synth.pair_np(9 + n, h, w, 3 * t, -t, blur=True)[0] for t in range(3)])
    dv = torch.from_numpy(video).to(gpu)
    assert np.array_equal(fm.process_sequence_device(dv).cpu().numpy(), fm.process_batch_device(dv[1:], dv[:-1]).cpu().numpy(), equal_nan=True)


@pytest.mark.parametrize("n", [120, 146, 60])
def test_half_tile_kernel_video_form(gpu, n):
    """r05: on a video the half-tile kernel keeps every frame's spectrum in the registers where the next pair's cross-power meets it
    (pc_half_kernel<CH, M, SEQ = true>: one image transform per pair instead of two; runs of consecutive pairs per workgroup, the run
    length chosen by the launcher). 38 frames = 37 pairs (more than one run at any run length the launcher picks for 4 patches), with a
    constant frame (constant boxes / degenerate pairs on both sides of it, the flags handed from `cur` to `prev`), a black frame and a
    repeated frame; every pair against the oracle at the bars of tests/tolerances.py, the same BITS as the pair entry on the same
    frames (the transform code is the same instantiation up to its sinks), and as a forced run length of 5 (runs that end mid-video)."""
    import tolerances
    gx, gy = 2, 2
    stride = (n // 2 + 3, n // 3 + 1)
    w, h = 5 + stride[0] * (gx - 1) + n + 2, 3 + stride[1] * (gy - 1) + n + 1
    F = 38
    video, _ = synth.video_torch(F, h, w, "cpu", k=n)
    video[7] = 93
    video[20] = 0
    video[30] = video[29]
    frames = video.numpy()
    fm = FftMethod(sample_point_size=n, frame_shape=(h, w), grid=(gx, gy), origin=(5, 3), stride=stride)
    assert fm.kernel_variant == "planned-half"
    dv = video.to(gpu)
    seq = fm.process_sequence_device(dv).cpu().numpy()
    pair = fm.process_batch_device(dv[1:], dv[:-1]).cpu().numpy()
    assert np.array_equal(seq, pair, equal_nan=True)
    lay = O.fft_layout(w, h, n, gx, gy, (5, 3), stride)
    checked = 0
    for k in range(F - 1):
        want64, _, diags = O.fft_process(frames[k + 1], frames[k], lay, 64, want_diag=True)
        want32, _ = O.fft_process(frames[k + 1], frames[k], lay, 32)
        for q in range(gx * gy):
            if np.isnan(want64[q]).any():
                assert np.isnan(seq[k][q]).all(), (k, q, seq[k][q])
            elif diags[q].second_value < 0.5 * diags[q].peak_value:
                checked += bool(tolerances.check_patch(seq[k][q], want64[q], want32[q], f"halfseq{n}/pair{k}", q,
                                                       pixels=tolerances.patch_pixels(frames[k + 1], frames[k], lay, q)))
    assert checked >= 0.8 * (F - 1) * gx * gy, checked
    import subprocess
    import sys
    code = ("import sys, numpy as np, torch; sys.path.insert(0, %r); from mrs_optic_flow_amd import FftMethod, synth;"
            "v, _ = synth.video_torch(%d, %d, %d, 'cpu', k=%d); v[7] = 93; v[20] = 0; v[30] = v[29];"
            "fm = FftMethod(sample_point_size=%d, frame_shape=(%d, %d), grid=(2, 2), origin=(5, 3), stride=%r);"
            "np.save(sys.argv[1], fm.process_sequence_device(v.to('cuda:0')).cpu().numpy())") % (ROOT, F, h, w, n, n, h, w, stride)
    out = os.path.join(ROOT, "gpurun_out", f"_halfseq_run5_{n}.npy")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    r = subprocess.run([sys.executable, "-c", code, out], env=dict(os.environ, MOF_FFT_SEQ_RUN="5"), capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    assert np.array_equal(np.load(out), seq, equal_nan=True)
    os.remove(out)


@pytest.mark.parametrize("n", [108, 49])
def test_planned_sizes_whose_video_form_is_the_half_tile_kernels(gpu, n):
    """Transform sizes 50, 54 and 108 keep the full-tile planned kernel for pairs (it is faster there) but run the half-tile kernel's
    sequence form on a video (one image transform per pair: +12 ... 28 %, profiles/r05_half_vs_planned_video.txt). The engine reports
    "planned"; the video entry is held to the oracle pair by pair (a constant frame and a repeated frame inside, more than one run), and it
    agrees with the pair entry -- another kernel family -- within 1e-4 px wherever both are held to 1e-4."""
    import tolerances
    gx, gy = 2, 2
    stride = (n // 2 + 3, n // 3 + 1)
    w, h = 5 + stride[0] * (gx - 1) + n + 2, 3 + stride[1] * (gy - 1) + n + 1
    F = 38
    video, _ = synth.video_torch(F, h, w, "cpu", k=n)
    video[7] = 93
    video[30] = video[29]
    frames = video.numpy()
    fm = FftMethod(sample_point_size=n, frame_shape=(h, w), grid=(gx, gy), origin=(5, 3), stride=stride)
    assert fm.kernel_variant == "planned"
    dv = video.to(gpu)
    seq = fm.process_sequence_device(dv).cpu().numpy()
    pair = fm.process_batch_device(dv[1:], dv[:-1]).cpu().numpy()
    assert not np.array_equal(seq, pair, equal_nan=True)  # (two kernel families: had the video entry run the pair kernel the bits would agree)
    assert np.array_equal(np.isnan(seq), np.isnan(pair))
    lay = O.fft_layout(w, h, n, gx, gy, (5, 3), stride)
    checked = 0
    for k in range(F - 1):
        want64, _, diags = O.fft_process(frames[k + 1], frames[k], lay, 64, want_diag=True)
        want32, _ = O.fft_process(frames[k + 1], frames[k], lay, 32)
        for q in range(gx * gy):
            if np.isnan(want64[q]).any():
                assert np.isnan(seq[k][q]).all(), (k, q, seq[k][q])
            elif diags[q].second_value < 0.5 * diags[q].peak_value:
                if tolerances.check_patch(seq[k][q], want64[q], want32[q], f"seqonly{n}/pair{k}", q, pixels=tolerances.patch_pixels(frames[k + 1], frames[k], lay, q)):
                    checked += 1
    assert checked >= 0.8 * (F - 1) * gx * gy, checked

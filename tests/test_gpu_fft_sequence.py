"""GPU tests of FftMethod's SEQUENCE mode (mof_fft_process_sequence_device, csrc/pc_seq_kernel.hip): a video on the
device, pair k = (frame k + 1, frame k) -- what consecutive processImage calls compute after `imPrev = imCurr.clone()`
(/root/reference/src/FftMethod.cpp:1872). 64 x 64 patches run the sequence kernel (one real transform per frame, the
previous spectrum in registers); the bar is the pair kernel's: 1e-4 px against the oracle on well-conditioned patches,
the same validity (NaN) pattern, and agreement with the pair kernel on (frames[1:], frames[:-1]) within 1e-4 px."""
import os

import numpy as np
import pytest
import torch

import oracle_lib as O
from mrs_optic_flow_amd import FftMethod, synth
from mrs_optic_flow_amd.engine import PEAK_OCL
from test_gpu_fft import TOL, _compare, _compare_ocl

pytestmark = pytest.mark.gpu


def _video_np(n_frames, h, w, k=0):
    frames, offs = synth.video_torch(n_frames, h, w, "cpu", k=k)
    return frames.numpy(), offs.numpy()


@pytest.mark.parametrize("shape,grid,origin,stride,n_frames", [
    ((480, 752), (8, 8), (1, 1), (98, 59), 7),      # BASELINE c2 layout
    ((448, 448), (7, 7), (0, 0), (64, 64), 5),      # the reference's own square tiling (sqNum = 7)
    ((70, 200), (2, 1), (3, 5), (129, 1), 38),      # 37 pairs: the 16-pair runs of a workgroup end twice
])
def test_video_matches_oracle_and_pair_kernel(gpu, shape, grid, origin, stride, n_frames):
    h, w = shape
    frames, offs = _video_np(n_frames, h, w, k=3)
    fm = FftMethod(sample_point_size=64, frame_shape=shape, grid=grid, origin=origin, stride=stride)
    video = torch.from_numpy(frames).to(gpu)
    got = fm.process_sequence_device(video)
    pairs = fm.process_batch_device(video[1:], video[:-1])
    torch.cuda.synchronize()
    got, pairs = got.cpu().numpy(), pairs.cpu().numpy()
    assert got.shape == (n_frames - 1, grid[0] * grid[1], 2)
    assert np.array_equal(np.isnan(got), np.isnan(pairs))
    assert np.allclose(got, pairs, rtol=0, atol=TOL, equal_nan=True)
    lay = O.fft_layout(w, h, 64, grid[0], grid[1], origin, stride)
    checked = 0
    for k in list(range(min(n_frames - 1, 4))) + ([15, 16, 17, 31, 32, 36] if n_frames > 37 else []):
        checked += _compare(got[k], frames[k + 1], frames[k], lay, f"pair{k}")
        # the planted motion: the window moved by (dox, doy) over the canvas = content moved by the negative of it
        d = -(offs[k + 1] - offs[k]).astype(float)
        assert np.allclose(np.nanmedian(got[k], axis=0), d, rtol=0, atol=0.5), (k, d)
    assert checked > 0.7 * min(n_frames - 1, 4) * grid[0] * grid[1]


def test_frames_in_a_wider_allocation(gpu):
    """Row pitch > width and frame stride > frame: the video is a view."""
    h, w, n = 130, 150, 9
    frames, _ = _video_np(n, h, w, k=1)
    big = torch.zeros((n, h + 5, w + 24), dtype=torch.uint8, device=gpu)
    big[:, 2:2 + h, 8:8 + w] = torch.from_numpy(frames).to(gpu)
    view = big[:, 2:2 + h, 8:8 + w]
    fm = FftMethod(sample_point_size=64, frame_shape=(h, w), grid=(2, 2), origin=(5, 1), stride=(81, 65))
    got = fm.process_sequence_device(view).cpu().numpy()
    dense = fm.process_sequence_device(torch.from_numpy(frames).to(gpu)).cpu().numpy()
    assert np.array_equal(got, dense, equal_nan=True)
    lay = O.fft_layout(w, h, 64, 2, 2, (5, 1), (81, 65))
    assert sum(_compare(got[k], frames[k + 1], frames[k], lay) for k in range(n - 1)) >= 24


def test_known_answers_and_degenerate_frames(gpu):
    """Circular shifts are exact; identical frames give 0; constant frames give the CPU path's degenerate answer
    (1 - N/2: a flat surface, first maximum at the corner -- tests/test_oracle_fft.py); a sequence of 0 or 1 frames has
    no pair."""
    n = 64
    gen = torch.Generator(device="cpu").manual_seed(11)
    base = torch.randint(0, 256, (n, n), dtype=torch.uint8, generator=gen)
    moves = [(3, -2), (-5, 7), (0, 0), (11, 0), (0, -9), (1, 1)]
    seq, pos = [base], (0, 0)
    for dx, dy in moves:
        pos = (pos[0] + dx, pos[1] + dy)
        seq.append(torch.roll(base, (pos[1], pos[0]), dims=(0, 1)))
    const = torch.full((n, n), 77, dtype=torch.uint8)
    video = torch.stack(seq + [const, const, base]).to(gpu)
    fm = FftMethod(n, n, 80.0)
    got = fm.process_sequence_device(video).cpu().numpy()[:, 0]
    assert np.allclose(got[:len(moves)], np.array(moves, float), rtol=0, atol=3e-5)
    assert not np.isnan(got[:len(moves)]).any()
    want = O.fft_process(const.numpy(), const.numpy(), O.fft_layout(n, n, n, 1, 1), 32)[0][0]
    assert np.allclose(got[len(moves) + 1], want, rtol=0, atol=TOL) and np.allclose(want, 1 - n / 2, atol=1e-4)
    assert fm.process_sequence_device(video[:1]).shape[0] == 0 and fm.process_sequence_device(video[:0]).shape[0] == 0


def test_ocl_peak_model_sequence(gpu):
    h = w = 128
    frames, _ = _video_np(5, h, w, k=5)
    fm = FftMethod(128, 64, 80.0, peak_model=PEAK_OCL)
    got = fm.process_sequence_device(torch.from_numpy(frames).to(gpu)).cpu().numpy()
    lay = O.fft_layout(w, h, 64, 2, 2, (0, 0), (64, 64))
    assert sum(_compare_ocl(got[k], frames[k + 1], frames[k], lay, 55, f"pair{k}") for k in range(4)) >= 12


def test_n128_video_matches_oracle_and_pair_kernel(gpu):
    """128 x 128 patches on a video: since r05 the half-tile kernel's video form (pc_half_kernel<CH, 128, SEQ>: c4seq 93 k -> 112 k pairs/s);
    MOF_FFT_SEQ_HALF128=1 keeps the older half-tile sequence kernel (pc_seq_half.hip), which the child-process test below re-runs this
    test on."""
    h, w, n, nf = 270, 480, 128, 21
    frames, offs = _video_np(nf, h, w, k=2)
    video = torch.from_numpy(frames).to(gpu)
    fm = FftMethod(sample_point_size=n, frame_shape=(h, w), grid=(3, 2), origin=(0, 0), stride=(119, 63))
    got = fm.process_sequence_device(video).cpu().numpy()
    pairs = fm.process_batch_device(video[1:], video[:-1]).cpu().numpy()
    assert np.array_equal(np.isnan(got), np.isnan(pairs))
    assert np.allclose(got, pairs, rtol=0, atol=TOL, equal_nan=True)
    lay = O.fft_layout(w, h, n, 3, 2, (0, 0), (119, 63))
    checked = 0
    for k in (0, 1, 15, 16, 17, 19):
        checked += _compare(got[k], frames[k + 1], frames[k], lay, f"pair{k}")
        d = -(offs[k + 1] - offs[k]).astype(float)
        assert np.allclose(np.nanmedian(got[k], axis=0), d, rtol=0, atol=0.5), (k, d)
    assert checked >= 30
    # circular shifts are exact; a constant pair gives the CPU path's degenerate answer
    gen = torch.Generator(device="cpu").manual_seed(3)
    base = torch.randint(0, 256, (n, n), dtype=torch.uint8, generator=gen)
    moves = [(5, -3), (-9, 14), (0, 0), (21, 1)]
    seq, pos = [base], (0, 0)
    for dx, dy in moves:
        pos = (pos[0] + dx, pos[1] + dy)
        seq.append(torch.roll(base, (pos[1], pos[0]), dims=(0, 1)))
    const = torch.full((n, n), 9, dtype=torch.uint8)
    one = FftMethod(n, n, 200.0)
    res = one.process_sequence_device(torch.stack(seq + [const, const]).to(gpu)).cpu().numpy()[:, 0]
    assert np.allclose(res[:len(moves)], np.array(moves, float), rtol=0, atol=5e-5)
    assert np.allclose(res[-1], 1 - n / 2, rtol=0, atol=1e-4)


def test_older_half_tile_sequence_kernel_at_128_stays_correct(gpu):
    """pc_seq_half.hip's sequence kernel (r03; the default for 128 x 128 videos until r05) is still in the library behind
    MOF_FFT_SEQ_HALF128=1: one child process re-runs the 128 x 128 video tests on it."""
    import subprocess
    import sys
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-k", "n128_video or (bgr_video and 128)",
                          "-p", "no:cacheprovider"], env=dict(os.environ, MOF_FFT_SEQ_HALF128="1"), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and " passed" in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]


def test_other_patch_sizes_run_the_pair_kernel_on_the_video(gpu):
    """(... or, r05, a video form with the pair entry's bits: 120 x 120 runs the half-tile kernel's)"""
    h, w, n = 250, 380, 120
    frames, _ = _video_np(4, h, w, k=2)
    video = torch.from_numpy(frames).to(gpu)
    fm = FftMethod(sample_point_size=n, frame_shape=(h, w), grid=(3, 2), origin=(3, 1), stride=(127, 129))
    assert torch.equal(fm.process_sequence_device(video), fm.process_batch_device(video[1:], video[:-1]), )


def test_full_size_c2seq_properties(gpu):
    """c2seq at BASELINE's size: 752 x 480, 8 x 8 patches of 64 x 64, 1025 frames. The video's window follows a closed
    path, so every pair has a known integer motion; every valid patch of every pair must sit within half a pixel of
    it, and the whole field must agree with the pair kernel."""
    h, w, n = 480, 752, 1025
    video, offs = synth.video_torch(n, h, w, gpu, k=0)
    fm = FftMethod(sample_point_size=64, frame_shape=(h, w), grid=(8, 8), origin=(1, 1), stride=(98, 59))
    got = fm.process_sequence_device(video)
    pairs = fm.process_batch_device(video[1:], video[:-1])
    torch.cuda.synchronize()
    assert torch.equal(torch.isnan(got), torch.isnan(pairs))
    assert torch.allclose(got, pairs, rtol=0, atol=TOL, equal_nan=True)
    d = -(offs[1:] - offs[:-1]).to(torch.float64).to(gpu)[:, None, :]
    ok = ~torch.isnan(got[..., 0])
    assert ok.float().mean() > 0.99
    assert ((got - d).abs().amax(dim=-1)[ok] < 0.5).all()


@pytest.mark.parametrize("n,fs", [(64, 192), (128, 256), (120, 240), (160, 320), (93, 186)])
def test_bgr_video_front_end(gpu, n, fs):
    """SURVEY N2 on the sequence path: a BGR8 video, crop + CV_RGB2GRAY (optic_flow.cpp:1609-1622) fused into the sequence
    kernels' loads (64: K1s; 128, 120, 160 and 93 -> 96: the half-tile kernel's video form, r05): identical bits to the gray entry on the
    converted crop, oracle bar."""
    nf, H, W, xi, yi = 5, fs + 9, fs + 24, 11, 5
    gray, _ = _video_np(nf, H, W, k=7)
    rng = np.random.default_rng(5)
    g = gray.astype(np.int32)
    col = np.clip(np.stack([g, 255 - g // 2, g * 3 // 4 + 20], axis=-1) + rng.integers(-2, 3, g.shape + (3,)), 0, 255).astype(np.uint8)
    fm = FftMethod(fs, n, 80.0)
    tv = torch.from_numpy(col).to(gpu)
    got = fm.process_sequence_device_bgr(tv[:, yi:yi + fs, xi:xi + fs]).cpu().numpy()
    conv = np.stack([O.rgb2gray(col[t, yi:yi + fs, xi:xi + fs]) for t in range(nf)])
    ref = fm.process_sequence_device(torch.from_numpy(conv).to(gpu)).cpu().numpy()
    assert np.array_equal(got, ref, equal_nan=True)
    lay = O.fft_layout(fs, fs, n, fs // n, fs // n)
    assert sum(_compare(got[k], conv[k + 1], conv[k], lay, f"bgr{k}") for k in range(nf - 1)) >= 0.7 * (nf - 1) * (fs // n) ** 2


def test_more_pairs_than_one_grid_dimension_holds(gpu):
    """The run index rides gridDim.z (65535 at most): with MOF_FFT_SEQ_RUN=1 a video of 65541 frames needs two launches.
    Run in a child process (the run length is read once per process). Circular shifts: every pair has a known answer."""
    import os
    import subprocess
    import sys
    code = r"""
import sys, numpy as np, torch
sys.path[:0] = [%r, %r]
from mrs_optic_flow_amd import FftMethod
n, F = 64, 65535 + 6
gen = torch.Generator(device="cpu").manual_seed(5)
base = torch.randint(0, 256, (n, n), dtype=torch.uint8, generator=gen)
moves = [(3, -2), (-5, 7), (0, 1), (11, 0)]
pos, P = (0, 0), []
for k in range(8):
    P.append(pos)
    pos = (pos[0] + moves[k %% 4][0], pos[1] + moves[k %% 4][1])
protos = torch.stack([torch.roll(base, (p[1], p[0]), dims=(0, 1)) for p in P]).cuda()
video = protos[torch.arange(F, device="cuda") %% 8]
got = FftMethod(n, n, 80.0).process_sequence_device(video).cpu().numpy()[:, 0]
step = np.array([(P[(k + 1) %% 8][0] - P[k][0], P[(k + 1) %% 8][1] - P[k][1]) for k in range(8)], float)
want = step[np.arange(F - 1) %% 8]
err = np.abs(got - want).max(axis=1)
print("ok" if err.max() < 5e-5 else ("bad", err.max(), int(err.argmax())))
""" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, MOF_FFT_SEQ_RUN="1"))
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.stdout[-500:], r.stderr[-1500:])

"""GPU tests added in round 2 for the gaps the round-1 review named:

* BASELINE c4 at its OWN geometry (1920x1080, 16x16 grid of 128x128 patches, origin (0,0), stride (119,63));
* the log-polar remap compared BYTE FOR BYTE with the oracle (cubic + Lanczos4, 240/256/480, both kernels);
* c5 with a batch that crosses the scale/rotation pipeline's chunk boundary;
* two batches on two different streams through one scale/rotation engine (engine-owned scratch);
* the HIP engine sharded over two ranks (gloo rendezvous, both ranks on the one GPU) == the 1-rank result, bit for bit;
* argument checking of the Python bindings (the C ABI only sees pointers and a pitch);
* the integer long-range gate (`int max_px_speed_lr`, include/FftMethod.h:393).
"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

import oracle_lib as O
import sr_scenes
from mrs_optic_flow_amd import FastSpacedBMMethod, FftMethod, ScaleRotationEstimator, synth
from mrs_optic_flow_amd.engine import INTER_CUBIC, INTER_LANCZOS4

pytestmark = pytest.mark.gpu
TOL = 1e-4
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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


# M = 30 at 480^2: source boxes beyond the staged kernel's 4 KB -> the table-in-LDS kernel serves the batch; M = 80: every
# ring class of the staged kernel down to one-dword boxes
@pytest.mark.parametrize("res,M", [(240, 40.0), (256, 45.0), (480, 49.9), (480, 30.0), (480, 80.0)])
@pytest.mark.parametrize("interp", [INTER_CUBIC, INTER_LANCZOS4])
@pytest.mark.parametrize("variant", [0, 1])   # cv::logPolar of OpenCV 4.x (Noetic) / cvLogPolar of OpenCV 3.2 (Melodic)
def test_logpolar_remap_is_byte_exact(gpu, res, M, interp, variant):
    """K4 against oracle_logpolar_u8, every byte, for both interpolations and both map variants (n < 4 images take the
    global-table kernel, n >= 4 the tile-stationary staged kernel: 5 images = one full and one ragged group of its
    register ring, 37 = four groups of eight and a ragged fifth); BORDER_TRANSPARENT pixels keep dst."""
    base = sr_scenes.canvas(5 + res + interp, res)
    frames = np.stack([sr_scenes.view(base, res, s, r) for s, r in [(1.0, 0.0), (1.05, 7.0), (0.93, -11.0),
                                                                   (1.0, 90.0), (1.2, 33.0)]])
    frames[4, :7, :] = 255  # saturating content next to the border (reflect-101 taps, clamping)
    est = ScaleRotationEstimator(res, M, logpolar_variant=variant)
    frames = frames[np.arange(37) % 5]
    frames[5:] = np.roll(frames[5:], 3, axis=2)  # not mere repeats
    big = torch.zeros((37, res + 2, res + 24), dtype=torch.uint8, device=gpu)
    big[:, 1:1 + res, 8:8 + res] = torch.from_numpy(frames).to(gpu)
    view = big[:, 1:1 + res, 8:8 + res]  # pitch > res, crop origin passed as the pointer
    fill = 37
    for n_img in (1, 5, 37):
        dst = torch.full((n_img, res, res), fill, dtype=torch.uint8, device=gpu)
        got = est.logpolar_batch_device(view[:n_img], interp, dst=dst).cpu().numpy()
        untouched = 0
        for k in range(n_img):
            want = O.logpolar(frames[k], M, interp, dst=np.full((res, res), fill, np.uint8), variant=variant)
            assert np.array_equal(got[k], want), (res, interp, n_img, k, int((got[k] != want).sum()))
            untouched += int((want == fill).sum())
        assert untouched > 0  # the outermost rings map outside the source: transparent pixels were exercised
    # a layout the staged kernel does not take (pitch and frame stride not multiples of 4): the table-in-LDS kernel
    odd = torch.zeros((6, res + 1, res + 7), dtype=torch.uint8, device=gpu)
    odd[:, 1:1 + res, 3:3 + res] = torch.from_numpy(frames[:6]).to(gpu)
    got = est.logpolar_batch_device(odd[:, 1:1 + res, 3:3 + res], interp).cpu().numpy()
    for k in range(6):
        assert np.array_equal(got[k], O.logpolar(frames[k], M, interp, variant=variant)), (res, interp, "odd pitch", k)
    zero = est.logpolar_batch_device(view[:1], interp).cpu().numpy()[0]  # default dst = zeros (tempIm, :27)
    assert np.array_equal(zero, O.logpolar(frames[0], M, interp, variant=variant))


@pytest.mark.parametrize("lanes", [1, 2])
def test_c5_batch_crossing_the_pipeline_chunk(gpu, lanes):
    """More pairs than one pass of the scale/rotation pipeline holds (mof_sr_config.batch_chunk, here 128 pairs; one
    stream lane and the two-lane remap / transform overlap): every pair, on both sides of the pass boundaries, equals
    the same pair processed alone, and samples match the oracle."""
    res, M, B = 240, 40.0, 300
    base = sr_scenes.canvas(91, res)
    protos = [(1.0, 0.0), (1.03, 2.0), (0.96, -3.0), (1.0, 5.0), (1.08, -1.0), (0.9, 8.0), (1.01, 0.5)]
    views = np.stack([sr_scenes.view(base, res, s, r) for s, r in protos])
    idx = np.arange(B) % len(protos)
    cur = torch.from_numpy(views[idx]).to(gpu)
    prev = torch.from_numpy(views[(idx * 3 + 1) % len(protos)]).to(gpu)
    est = ScaleRotationEstimator(res, M, batch_chunk=128, pipeline_lanes=lanes)
    got = est.process_batch_device(cur, prev)
    torch.cuda.synchronize()
    got = got.cpu().numpy()
    for k in (0, 15, 16, 31, 32, 63, 64, 65, 127, 128, 129, 149, 254, 255, 256, 257, 299):
        alone = est.process_batch_device(cur[k:k + 1], prev[k:k + 1]).cpu().numpy()[0]
        assert np.array_equal(alone, got[k]), k
    # the (cur, prev) prototypes of pair k depend on k mod 7 only: identical bits wherever a pair sits in the batch
    for k in range(len(protos), B):
        assert np.array_equal(got[k], got[k % len(protos)]), k
    for k in (3, 64, 255, 256, 299):
        ref = O.ScaleRotationEstimator(res, M, 64)
        ref.processImage(views[(idx[k] * 3 + 1) % len(protos)])
        s, r = ref.processImage(views[idx[k]])
        assert np.allclose(got[k, 2:], ref.pt, rtol=0, atol=TOL), (k, got[k], ref.pt)
        assert abs(got[k, 0] - s) < 1e-5 and abs(got[k, 1] - r) < 1e-5


def test_scale_rotation_engine_on_two_streams(gpu):
    """The scale/rotation pipeline runs through engine-owned scratch: batches issued back to back on two different
    streams must not corrupt each other (the second stream waits for the first batch's last kernel)."""
    res, M, B = 256, 45.0, 48
    base = sr_scenes.canvas(17, res)
    a = np.stack([sr_scenes.view(base, res, 1.0 + 0.01 * (k % 5), 1.5 * (k % 7)) for k in range(B)])
    b = np.stack([sr_scenes.view(base, res, 1.0 - 0.01 * (k % 4), -2.0 * (k % 3)) for k in range(B)])
    ta, tb = torch.from_numpy(a).to(gpu), torch.from_numpy(b).to(gpu)
    est = ScaleRotationEstimator(res, M)
    want1 = est.process_batch_device(ta, tb).clone()
    want2 = est.process_batch_device(tb, ta).clone()
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for _ in range(4):
        with torch.cuda.stream(s1):
            got1 = est.process_batch_device(ta, tb)
        with torch.cuda.stream(s2):
            got2 = est.process_batch_device(tb, ta)
        with torch.cuda.stream(s1):
            got3 = est.process_batch_device(ta, tb)
        torch.cuda.synchronize()
        assert torch.equal(got1, want1) and torch.equal(got2, want2) and torch.equal(got3, want1)
    # the stateful entry (engine's own stream) right behind a batch on another stream
    with torch.cuda.stream(s2):
        got2 = est.process_batch_device(tb, ta)
    est.reset()
    est.processImage(b[0])
    s, r = est.processImage(a[0])
    torch.cuda.synchronize()
    assert torch.equal(got2, want2)
    # (the batch entry runs every pair through the kernels of the stateful entry: a fresh estimator fed (prev, cur) -- same bits)
    assert (s, r) == (float(want1[0, 0]), float(want1[0, 1]))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


_RANK_SCRIPT = r"""
import os, sys
sys.path[:0] = [{root!r}, os.path.join({root!r}, "tests")]
import numpy as np, torch, torch.distributed as dist
from mrs_optic_flow_amd import FftMethod, FastSpacedBMMethod, sharding, synth
rank, world, port, n_pairs, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{{port}}", rank=rank, world_size=world)
dev = torch.device("cuda", 0)   # both ranks share the one GPU of the box
h, w, n = 480, 752, 64
fm = FftMethod(sample_point_size=n, frame_shape=(h, w), grid=(8, 8), origin=(1, 1), stride=(98, 59))
bm = FastSpacedBMMethod(16, 16, 8, (h, w))
def fft_shard(lo, hi):
    cur, prev, _, _ = synth.batch_torch(hi - lo, h, w, n // 8, dev, k0=lo)
    r = fm.process_batch_device(cur, prev); torch.cuda.synchronize(); return r.cpu()
def bm_shard(lo, hi):
    cur, prev, _, _ = synth.batch_torch(hi - lo, h, w, 12, dev, k0=lo)
    dx, dy, mode = bm.process_batch_device(cur, prev); torch.cuda.synchronize()
    return torch.cat([dx.reshape(hi - lo, -1), dy.reshape(hi - lo, -1), mode], dim=1).cpu()
full = sharding.run_sharded(fft_shard, n_pairs, rank, world)
full_bm = sharding.run_sharded(bm_shard, n_pairs, rank, world)
np.savez(out, fft=full.numpy(), bm=full_bm.numpy())
dist.barrier(); dist.destroy_process_group()
"""


@pytest.mark.parametrize("n_pairs", [37])
def test_two_rank_hip_engine_equals_single_rank(gpu, tmp_path, n_pairs):
    """the batch sharded over two ranks (fresh processes, gloo rendezvous on 127.0.0.1, both on the one
    GPU) and gathered with sharding.run_sharded gives the 1-rank result bit for bit -- c2 geometry for the FFT path,
    c3 geometry for the block scan; 37 pairs -> shards of 19 and 18 (ragged)."""
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT.format(root=ROOT))
    port = _free_port()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "2", str(port), str(n_pairs),
                               str(tmp_path / f"r{r}.npz")], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                              text=True) for r in range(2)]
    for p in procs:
        out, _ = p.communicate(timeout=600)
        assert p.returncode == 0, out[-3000:]
    h, w, n = 480, 752, 64
    fm = FftMethod(sample_point_size=n, frame_shape=(h, w), grid=(8, 8), origin=(1, 1), stride=(98, 59))
    cur, prev, _, _ = synth.batch_torch(n_pairs, h, w, n // 8, gpu, k0=0)
    want = fm.process_batch_device(cur, prev).cpu().numpy()
    bm = FastSpacedBMMethod(16, 16, 8, (h, w))
    cur, prev, _, _ = synth.batch_torch(n_pairs, h, w, 12, gpu, k0=0)
    dx, dy, mode = bm.process_batch_device(cur, prev)
    want_bm = torch.cat([dx.reshape(n_pairs, -1), dy.reshape(n_pairs, -1), mode], dim=1).cpu().numpy()
    for r in range(2):
        got = np.load(tmp_path / f"r{r}.npz")
        assert got["fft"].shape == want.shape and np.array_equal(got["fft"], want, equal_nan=True)
        assert np.array_equal(got["bm"], want_bm)


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


def test_multi_pass_batch_is_graph_capturable(gpu):
    """A scale/rotation batch longer than one pipeline pass, on one lane and as two lanes (pipeline_lanes = 2: remaps on
    the engine's own stream beside the transforms). Under HIP-graph capture the engine's stream joins the caller's capture
    by an event fork / join and the graph must replay to the bits of the eager run. Run in a child process (batch_chunk =
    2: seven pairs = four passes) at 240 and 480."""
    script = os.path.join(ROOT, "tools", "check_graph_capture.py")
    for res in ("240", "480"):
        r = subprocess.run([sys.executable, script, res], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and f"graph ok {res}" in r.stdout, (r.stdout + r.stderr)[-2000:]

"""Round-4 GPU tests: bench.py launching its own ranks, the native (C ABI) sharded batch entry, graph-lifetime advisor case,
seeded regression classes from the fuzzers."""
import gc
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import oracle_lib as O
import tolerances
from mrs_optic_flow_amd import FftMethod, MofError, ScaleRotationEstimator, release_captured, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 1e-4


def test_bench_launches_its_own_ranks(gpu):
    """`python bench.py --gpus 2 ...` run BARE (no RANK / WORLD_SIZE in the environment -- the shape of the driver's command): the
    parent touches no GPU, starts two fresh rank processes, relays rank 0's JSON line and exits 0. (gloo + --share-gpu: two
    ranks rehearse on the one GPU of this box; with RCCL the same code path needs N GPUs.)"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-gpu",
                        "--workload", "c2", "--batch", "16", "--steps", "3", "--warmup", "1", "--sustain-s", "0",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # only rank 0 prints, once
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["scaling"] == "weak" and line["config"]["batch_per_gpu"] == 16
    # a failing rank fails the launcher (an impossible workload argument makes argparse exit 2 in every rank)
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-gpu",
                          "--workload", "c2", "--batch", "-1", "--steps", "1", "--warmup", "0", "--sustain-s", "0",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert bad.returncode != 0


def test_release_captured_of_one_engine_leaves_other_graphs_replayable(gpu):
    """Advisor r03: release_captured(B) used to purge the process-wide parked list, freeing engine A -- closed while graph A could
    still replay -- under graph A. Two graphs; close A's engine; release B; replay A."""
    from mrs_optic_flow_amd import _capi
    from mrs_optic_flow_amd import engine as E

    lib = _capi.load()
    B, fs = 4, 256
    cur, prev, _, _ = synth.batch_np(B, fs, fs, 6, classes=False, k0=21)
    tc, tp = torch.from_numpy(cur).to(gpu), torch.from_numpy(prev).to(gpu)
    fa, fb = FftMethod(fs, 64, 80.0), FftMethod(fs, 128, 80.0)
    want_a = fa.process_batch_device(tc, tp).clone()
    torch.cuda.synchronize()
    parked0 = lib.mof_deferred_count()
    ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        with torch.cuda.graph(ga, stream=side):
            out_a = fa.process_batch_device(tc, tp)
        with torch.cuda.graph(gb, stream=side):
            out_b = fb.process_batch_device(tc, tp)
    # close engine A while graph A lives: the library parks it (Python's keep-alive set bypassed, as a C++ host would)
    E._CAPTURED.discard(fa)
    del fa
    gc.collect()
    assert lib.mof_deferred_count() == parked0 + 1
    del gb
    assert release_captured(fb) == 1              # B's graphs are gone ...
    assert lib.mof_deferred_count() == parked0 + 1  # ... which says nothing about A: still parked, not freed
    out_a.zero_()
    ga.replay()
    torch.cuda.synchronize()
    assert torch.equal(out_a, want_a)
    del ga, out_b
    release_captured()                            # every graph is gone: now the parked engines are freed
    assert lib.mof_deferred_count() == 0


def test_native_sharded_entry_from_a_cpp_host(gpu):
    """tests/cpp/test_shard.cpp: mof_shard_fft_* and (r05) mof_shard_bm_* -- one process, one engine and stream per device,
    ceil(B / G) contiguous shards, ONE in-place RCCL all-gather (explicit mof_shard_*_init_gather = ncclCommInitAll, then
    ncclAllGather through the run-time-bound librccl) -- with the devices this box has; every device's gathered result equals the
    single-engine result on the whole batch bit for bit (FFT vectors; block shifts and modes in one slab)."""
    binp = os.path.join(ROOT, "tests", "cpp", "test_shard")
    assert os.path.exists(binp), "tests/cpp/test_shard missing: run __graft_entry__.build()"
    for pairs in (37, 8):
        r = subprocess.run([binp, str(pairs)], capture_output=True, text=True, timeout=300,
                           env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
        assert r.returncode == 0 and f"shard ok {torch.cuda.device_count()} {pairs}" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


def test_native_sharded_entry_through_ctypes(gpu):
    """The same entry from Python's ctypes binding, without the gather (gather = 0 needs no RCCL) and with it."""
    import ctypes as C
    from mrs_optic_flow_amd import _capi

    lib = _capi.load()
    B, h, w = 11, 136, 200
    cur, prev, _, _ = synth.batch_np(B, h, w, 5, k0=9)
    fm = FftMethod(sample_point_size=64, frame_shape=(h, w), grid=(2, 2), origin=(3, 1), stride=(97, 59))
    tc, tp = torch.from_numpy(cur).to(gpu), torch.from_numpy(prev).to(gpu)
    want = fm.process_batch_device(tc, tp)
    grp = C.c_void_p()
    _capi.check(lib.mof_shard_fft_create(C.byref(fm.cfg), None, 1, C.byref(grp)))
    try:
        assert lib.mof_shard_fft_devices(grp) == 1
        for gather in (0, 1):
            out = torch.full((B, 4, 2), float("nan"), dtype=torch.float64, device=gpu)
            pc, pp, po = (C.c_void_p * 1)(tc.data_ptr()), (C.c_void_p * 1)(tp.data_ptr()), (C.c_void_p * 1)(out.data_ptr())
            torch.cuda.synchronize()
            if gather:  # r05: the gather's set-up is explicit -- the asynchronous call refuses to build communicators itself
                assert lib.mof_shard_fft_process_batch_device(grp, pc, tc.stride(0), pp, tp.stride(0), tc.stride(1), B, po, 1) == _capi.MOF_ERR_NOT_INIT
                _capi.check(lib.mof_shard_fft_init_gather(grp))
                assert lib.mof_shard_fft_gather_ready(grp) == 1
            _capi.check(lib.mof_shard_fft_process_batch_device(grp, pc, tc.stride(0), pp, tp.stride(0), tc.stride(1), B, po, gather))
            _capi.check(lib.mof_shard_fft_sync(grp))
            assert torch.equal(out, want), gather
    finally:
        lib.mof_shard_fft_destroy(grp)


# ---- the fuzzers' input classes, seeded, through EVERY entry point ----------------------------------------
_DBL_EPS, _FLT_EPS = float(np.finfo(np.float64).eps), float(np.finfo(np.float32).eps)


def _expected(cur, prev, lay, max_speed=80.0):
    """Per patch: (want [2] or None, tolerance). Constant patches: the closed form of cv::phaseCorrelate's flat surface (first
    index, clamped 3 x 3 centroid of equal values: 9c / (9c + DBL_EPSILON) - M/2 with c = C_dc = P / (P^2 + FLT_EPSILON), P the
    product of the two pixel sums) -- the oracle's own radix-3/5 transform of a constant is not exactly zero off DC, OpenCV's
    neither, so the closed form (what exact arithmetic gives) is the bar there. With zero padding (M > N) only an all-zero patch
    stays constant. Everything else: the oracle, where its arg-max is stable (tests/test_gpu_fft.py::_compare)."""
    n, gx, gy = lay.patch, lay.grid_x, lay.grid_y
    m = O.optimal_dft_size(n)
    want64, _, diags = O.fft_process(cur, prev, lay, 64, want_diag=True)
    want32, _ = O.fft_process(cur, prev, lay, 32)
    out = []
    for j in range(gy):
        for i in range(gx):
            p = i + j * gx
            x0, y0 = lay.origin_x + i * lay.stride_x, lay.origin_y + j * lay.stride_y
            a, b = cur[y0:y0 + n, x0:x0 + n], prev[y0:y0 + n, x0:x0 + n]
            ca, cb = int(a.max()) == int(a.min()), int(b.max()) == int(b.min())
            deg = (ca or cb) if m == n else ((ca and a.max() == 0) or (cb and b.max() == 0))
            if deg:
                P = float(a.astype(np.float64).sum()) * float(b.astype(np.float64).sum())
                c9 = 9.0 * P / (P * P + _FLT_EPS) if P > 0 else 0.0
                s = (c9 / (c9 + _DBL_EPS) if c9 > 0 else 0.0) - m / 2.0
                bad = 2 * s * s > max_speed ** 2 or abs(s) > n / 2.0
                out.append((np.array([np.nan, np.nan]) if bad else np.array([s, s]), 1e-4))
                continue
            # only a CLEAR peak pins the answer: on unrelated or flat-against-texture content (a constant patch that zero padding
            # turned into a box; a frame next to an unrelated one in the video below) the surface is noise, its arg-max and its
            # near-cancelling centroid are decided by rounding, and the two oracle precisions agreeing with each other (same
            # algorithm, same order of operations) says nothing about a third arithmetic
            if not diags[p].second_value < 0.5 * diags[p].peak_value:
                out.append((None, 0.0))
                continue
            out.append(((want64[p], want32[p]), TOL))  # (both oracles: the bars of tests/tolerances.py, f32-limited patches included)
    return out


def _check(got, cur, prev, lay, label):
    n_checked = 0
    for p, (want, tol) in enumerate(_expected(cur, prev, lay)):
        if want is None:
            continue
        if isinstance(want, tuple):
            n_checked += bool(tolerances.check_patch(got[p], want[0], want[1], label, p))
            continue
        assert np.allclose(got[p], want, rtol=0, atol=tol, equal_nan=True), (label, p, got[p], want)
        n_checked += 1
    return n_checked


@pytest.mark.parametrize("n", [32, 64, 120, 128, 60, 62, 160])
def test_fuzzer_classes_through_every_entry_point(gpu, n):
    """one-sided constant frames, black frames, a constant rectangle inside a frame, a saturated region, strongly low-passed
    content and exactly-cancelling alternating sums -- through the pair batch, the BGR front end, the sequence entry, the
    stateful processImage and the long-range mode, at tuned (32 / 64 / 120 / 128), planned (60, 62 -> 64) and large (160) sizes."""
    fs = 2 * n  # 2 x 2 patches, reference tiling
    classes = synth.fuzz_classes_np(100 + n, fs, fs, 3, -2)
    lay = O.fft_layout(fs, fs, n, 2, 2)
    fm = FftMethod(fs, n, 80.0)
    names = sorted(classes)
    cur = np.stack([classes[k][0] for k in names])
    prev = np.stack([classes[k][1] for k in names])
    tc, tp = torch.from_numpy(cur).to(gpu), torch.from_numpy(prev).to(gpu)
    # (1) independent pairs
    got = fm.process_batch_device(tc, tp).cpu().numpy()
    total = 0
    for k, name in enumerate(names):
        total += _check(got[k], cur[k], prev[k], lay, f"n{n}/pair/{name}")
    assert total >= 0.6 * 4 * len(names), total
    # (2) the same frames as interleaved BGR8 with B = G = R (CV_RGB2GRAY then returns the value itself): same bits
    bgr_c, bgr_p = tc[..., None].expand(-1, -1, -1, 3).contiguous(), tp[..., None].expand(-1, -1, -1, 3).contiguous()
    assert np.array_equal(fm.process_batch_device_bgr(bgr_c, bgr_p).cpu().numpy(), got, equal_nan=True)
    # (3) a video that walks through the classes: texture, constant, texture, black, texture with a constant rectangle, smooth ...
    tex_c, tex_p = synth.pair_np(7 + n, fs, fs, 2, 1)
    video = np.stack([tex_p, tex_c, classes["const_cur"][0], tex_c, classes["black_both"][0], classes["const_rect"][0], tex_p,
                      classes["smooth"][1], classes["smooth"][0], classes["saturated"][0], classes["checker"][1], classes["checker"][0]])
    seq = fm.process_sequence_device(torch.from_numpy(video).to(gpu)).cpu().numpy()
    for k in range(len(video) - 1):
        _check(seq[k], video[k + 1], video[k], lay, f"n{n}/seq/{k}")
    # (4) the stateful entry, frame by frame (first frame against itself)
    fm.reset()
    for k in range(len(video)):
        out = fm.processImage(video[k])
        _check(out, video[k], video[k - 1] if k else video[k], lay, f"n{n}/stateful/{k}")
    # (5) long-range mode: frames whose quarter-resolution reduction IS the class frame (every 4 x 4 cell constant), one patch
    if n <= 128:
        flr = FftMethod(4 * n, n, 80.0)
        lay1 = O.fft_layout(n, n, n, 1, 1)
        sel = [k for k, name in enumerate(names)]
        small_c = np.stack([cur[k][:n, :n] for k in sel])
        small_p = np.stack([prev[k][:n, :n] for k in sel])
        up = lambda a: np.repeat(np.repeat(a, 4, axis=1), 4, axis=2)
        lr = flr.process_long_range_batch_device(torch.from_numpy(up(small_c)).to(gpu), torch.from_numpy(up(small_p)).to(gpu)).cpu().numpy()
        for k in sel:
            _check(lr[k], small_c[k], small_p[k], lay1, f"n{n}/longrange/{names[k]}")


def test_mfma_first_stage_variant_of_k1_passes_the_parity_tests(gpu):
    """K1 (N = 64) with S1 of its forward transform on the matrix cores (pc_passes3.hpp, fwd3_rows_mfma;
    f16 hi + lo split of the DFT-16 matrix, f32 accumulation) is an A/B library (`make mfma`), measured slower than the product
    (profiles/r04_mfma_s1_ab.txt) and therefore not shipped -- but it has to stay correct for that comparison to mean anything:
    a child process re-runs the N = 64 parity cases of test_gpu_fft.py on it."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "mrs_optic_flow_amd", "csrc", "ab", "libmof_hip_mfma.so")
    if not os.path.exists(lib):
        pytest.skip("csrc/ab/libmof_hip_mfma.so not built (`make -C mrs_optic_flow_amd/csrc mfma`)")
    env = dict(os.environ, MOF_LIB_PATH=lib)
    sel = "golden or seeded or ocl_peak or bgr or long_range or gating or circular or expected_variant"
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_fft.py"), "-q", "-x", "-k", sel,
                          "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert " passed" in out.stdout


def test_fused_estimator_kernel_passes_the_estimator_parity_tests(gpu):
    """K56 (sr_fused_kernel.hip, MOF_SR_FUSED=1): the estimator's row transforms as a dense product on the matrix cores inside the
    column kernel -- no row spectra in HBM (5.1 GB -> 1.4 GB per 1024-pair pass of c5), but measured slower than K5s + K6s
    (profiles/r04_sr_fused_*), so it is an opt-in path of the batch entry. It has to stay correct: a child process re-runs the
    estimator's parity tests (all three tuned resolutions, golden vectors, full-size c5 case) with the knob set."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MOF_SR_FUSED="1")
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_sr.py"),
                          os.path.join(root, "tests", "test_gpu_r03.py"), "-q", "-x", "-m", "gpu", "-k", "not stateful and not sequence",
                          "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert " passed" in out.stdout


def test_patch_with_an_exactly_zero_spectral_bin(gpu):
    """Found by tools/fft_sr_fuzz.py (seed 20261004, trial 28): an ordinary textured 120 x 120 patch whose PREVIOUS image has a bin
    that is zero in exact arithmetic (at (N/3, 2N/3) the DFT is S0 + S1 w + S2 w^2 over the residue classes of y + 2x mod 3, and
    the three integer sums happen to be equal). The f64 oracle gets 1e-13 there and the f32 oracle's radix order cancels exactly
    too, so the two agree to 1e-5 px -- but any other f32 transform (the tuned kernel, the planned kernel, numpy's) leaves
    1e-7-relative noise, the cross-power normalisation turns it into a unit-magnitude bin, and the centroid moves by 2e-4 px.
    That is the f32-limited class of DESIGN "K1 planned / Tolerances" in a form the oracle-to-oracle distance does not show:
    oracle_lib.f32_floor_bins detects it, and the bar there is 1e-4 + 2 bins / (M^2 x normalised peak), at most 1e-3 px (tests/tolerances.py)."""
    n, (gx, gy), (ox, oy), (sx, sy), (h, w), k0 = 120, (4, 4), (6, 2), (95, 153), (589, 417), 897
    cur, prev, _, _ = synth.batch_np(3, h, w, 15, k0=k0)
    lay = O.fft_layout(w, h, n, gx, gy, (ox, oy), (sx, sy))
    k, p = 1, 6
    x0, y0 = ox + (p % gx) * sx, oy + (p // gx) * sy
    a, b = cur[k][y0:y0 + n, x0:x0 + n], prev[k][y0:y0 + n, x0:x0 + n]
    bins = O.f32_floor_bins(a, b)
    assert bins == 2, bins  # the bin and its mirror, of prev
    assert O.f32_floor_bins(cur[0][y0:y0 + n, x0:x0 + n], prev[0][y0:y0 + n, x0:x0 + n]) == 0
    want64, _, diags = O.fft_process(cur[k], prev[k], lay, 64, want_diag=True)
    want32, _ = O.fft_process(cur[k], prev[k], lay, 32)
    assert np.abs(want64[p] - want32[p]).max() < 2e-5  # the oracles do not see it
    slack = tolerances.floor_bins_bar(bins, diags[p].peak_value)  # 1e-4 + 2 bins / peak, never above 1e-3 px
    fm = FftMethod(sample_point_size=n, frame_shape=(h, w), grid=(gx, gy), origin=(ox, oy), stride=(sx, sy))
    got = fm.process_batch_device(torch.from_numpy(cur).to(gpu), torch.from_numpy(prev).to(gpu)).cpu().numpy()[k]
    others = np.delete(np.arange(gx * gy), p)
    assert np.abs(got[others] - want64[others]).max() < 1e-4
    e32, e64 = float(np.abs(got[p] - want32[p]).max()), float(np.abs(got[p] - want64[p]).max())
    tolerances.record_floor_bins("n120/exact-zero-bin", p, bins, slack, e32, e64)
    assert e32 <= slack and e64 <= slack, (got[p], want64[p], slack)


@pytest.mark.parametrize("case", ["in_lds_118", "in_lds_124_odd", "large_158", "large_146", "half_156", "half_152", "half_56", "large_232", "in_lds_130_odd"])
def test_constant_frame_against_texture_on_padded_patches(gpu, case):
    """Found by tools/fft_sr_fuzz.py's sequence trials (seeds 101 / 202): ONE frame of the pair constant, patch size below its
    transform size. cv::phaseCorrelate pads the constant patch to an n x n box whose spectrum is level x D[v] D[u], exactly zero on
    the Nyquist lines. (a) large-patch pipeline: rows are transformed in pairs, the spectra of rows 2j and 2j + 1 of a constant
    image differ by rounding and their alternating column sum is 79 x that instead of 0 -- 0.04 px off; L6 now zeroes those
    bins from L5's flags (box_zeros). (b) in-LDS planned kernel: the packed transform delivers the box with the textured
    patch's rounding noise on top -- 1e-3 px off; the kernel now takes the box from its closed form (D in LDS) and the textured
    spectrum as Z -+ i box (D summed in f64: 124 f32 additions lose 4e-4 of it, 1e-3 px on the 124 -> 125 case). Both the pair entry
    and the sequence entry, against the bars of tests/tolerances.py. (r05: 146 and 158 pad to 150 / 160 and run the fused half-tile kernel, which applies the same box_zeros rule.)"""
    n, grid, origin, stride, (h, w), k, const = {
        "in_lds_118": (118, (3, 3), (4, 5), (76, 84), (294, 281), 68, (0, 120)),
        "in_lds_124_odd": (124, (1, 2), (7, 8), (136, 123), (256, 136), 988, (0, 39)),  # pads to 125: no Nyquist lines, |box bin| = level everywhere
        "large_158": (158, (1, 2), (3, 2), (154, 169), (331, 166), 777, (1, 169)),
        "large_146": (146, (1, 2), (4, 2), (53, 156), (308, 156), 634, (0, 84)),
        # r05 (tools/fft_sr_fuzz.py seeds 606 / 608 at 160 trials): the box's EXACT zeros are all the lines k != 0 with k n = 0 (mod M) -- the
        # multiples of M / gcd(n, M) --, not the Nyquist line alone: 156 in 160 -> 40, 80, 120; 152 in 160 -> every multiple of 20; 56 in 60 ->
        # 15, 30, 45; 232 in 240 -> multiples of 30; 130 in 135 (odd M) -> multiples of 27. 0.03 - 0.09 px off before the rule was generalised
        # (csrc/pc_common.hpp: box_zero_period) in all three kernel families.
        "half_156": (156, (1, 2), (1, 3), (53, 79), (249, 167), 611, (1, 160)),
        "half_152": (152, (1, 1), (7, 0), (83, 150), (158, 167), 612, (0, 212)),
        "half_56": (56, (2, 2), (3, 4), (60, 58), (125, 127), 613, (0, 97)),
        "large_232": (232, (1, 1), (2, 3), (1, 1), (240, 238), 614, (1, 55)),
        "in_lds_130_odd": (130, (1, 1), (2, 2), (1, 1), (136, 134), 615, (0, 201)),
    }[case]
    video, _ = synth.video_torch(2, h, w, "cpu", k=k)
    video[const[0]] = const[1]
    frames = video.numpy()
    fm = FftMethod(sample_point_size=n, frame_shape=(h, w), grid=grid, origin=origin, stride=stride)
    mm, half_on = O.optimal_dft_size(n), os.environ.get("MOF_FFT_HALF", "") != "0"
    want_variant = ("planned-half" if half_on and mm in (60, 72, 90, 96, 100, 120, 144, 150, 160, 162, 180, 192) else ("planned" if mm <= 135 else "planned-large"))
    assert fm.kernel_variant == want_variant
    dv = video.to(gpu)
    pair = fm.process_batch_device(dv[1:], dv[:-1]).cpu().numpy()[0]
    seq = fm.process_sequence_device(dv).cpu().numpy()[0]
    lay = O.fft_layout(w, h, n, grid[0], grid[1], origin, stride)
    want64, _, diags = O.fft_process(frames[1], frames[0], lay, 64, want_diag=True)
    want32, _ = O.fft_process(frames[1], frames[0], lay, 32)
    checked = 0
    for p in range(want64.shape[0]):
        if not diags[p].second_value < 0.5 * diags[p].peak_value:
            continue
        dd = float(np.abs(want64[p] - want32[p]).max())
        if case == "in_lds_130_odd":
            # an ODD zero period (27): the f64 oracle gets ~1e-13 in the box's zero bins (C = 0, the exact-arithmetic answer), the f32 oracle's
            # radix-3/5 sums leave 1e-7-relative noise there that the normalisation turns into eight lines of unit-magnitude bins: the two
            # oracles are 0.065 px apart. The kernel knows the box exactly and zeroes those lines: it must give the exact-arithmetic answer.
            assert dd > 1e-3, (case, p, dd)
            assert np.abs(pair[p] - want64[p]).max() <= 1e-4 and np.abs(seq[p] - want64[p]).max() <= 1e-4, (case, p, pair[p], seq[p], want64[p])
            checked += 1
            continue
        assert dd < 2e-4, (case, p, dd)
        checked += 1
        tolerances.check_patch(pair[p], want64[p], want32[p], case + "/pair", p)  # (the bars of tests/tolerances.py)
        tolerances.check_patch(seq[p], want64[p], want32[p], case + "/seq", p)
    assert checked >= want64.shape[0] - 1, (case, checked)


def test_split_lane_column_kernel_passes_the_estimator_parity_tests(gpu):
    """K6p (sr_seq_kernel.hip: sr_cols_split_kernel, MOF_SR_COLS_SPLIT=1, resolution 480): two columns per wave, the radix-32 stage of a
    column transform split over lane pairs, one LDS round trip per transform, three waves per SIMD -- 558 against 585 us per 1024 pairs,
    c5 unchanged within the noise (profiles/r04_k6p_ab.txt), so it is an opt-in form. A child process re-runs the estimator's parity
    tests with the knob, incl. the bit-identity of a sequence with its frame-by-frame stateful calls."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MOF_SR_COLS_SPLIT="1")
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_sr.py"),
                          os.path.join(root, "tests", "test_gpu_sr_sequence.py"), "-q", "-x", "-m", "gpu", "-k", "480 or golden or batch",
                          "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert " passed" in out.stdout

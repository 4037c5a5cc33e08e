"""Round-5 GPU tests: the native shard group with G > 1 on the one-GPU box (rehearsal knob), block matching through the shard group
(dx, dy and mode in one slab), and the round's new kernel forms."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
import torch

import oracle_lib as O
from mrs_optic_flow_amd import FastSpacedBMMethod, FftMethod, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 1e-4


@pytest.mark.parametrize("pairs,G", [(37, 2), (37, 4), (9, 4), (3, 4), (1, 2), (8, 2)])
def test_shard_group_with_several_shards_on_one_device(gpu, pairs, G):
    """`mof_shard_*_process_batch_device`'s G > 1 arithmetic -- slab i at i * slab, ragged last shards
    (37 over 4 -> 10 10 10 7), empty ones (9 over 4 -> 3 3 3 0; 3 over 4 -> 1 1 1 0; 1 over 2) -- executed for real: the rehearsal knob
    MOF_SHARD_SHARE_DEVICE=1 admits G shards on the one device with gather = 0, and tests/cpp/test_shard.cpp checks that every
    slab lands at its place bit-equal to the single-engine call and that nothing else of the buffer is written -- FftMethod vectors,
    and FastSpacedBMMethod's dx | dy | mode planes. The all-gather itself stays a 1-rank run (RCCL: one rank per device) until a
    multi-GPU node exists; a shared-device group refuses it (checked inside)."""
    binp = os.path.join(ROOT, "tests", "cpp", "test_shard")
    assert os.path.exists(binp), "tests/cpp/test_shard missing: run __graft_entry__.build()"
    r = subprocess.run([binp, "rehearse", str(pairs), str(G)], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, MOF_SHARD_SHARE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0 and f"rehearse ok {G} {pairs}" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


def test_block_matching_shard_group_through_ctypes(gpu):
    """mof_shard_bm_* from the ctypes binding on a one-device group, with the (1-rank) RCCL gather: per-block shifts and the
    per-pair modes come back from ONE slab ("BM mode vectors ride in the same slab"), bit-equal to the engine's
    own batch call."""
    from mrs_optic_flow_amd import _capi

    lib = _capi.load()
    B, h, w = 7, 136, 200
    cur, prev, _, _ = synth.batch_np(B, h, w, 5, k0=3)
    bm = FastSpacedBMMethod(16, 8, 8, (h, w))
    tc, tp = torch.from_numpy(cur).to(gpu), torch.from_numpy(prev).to(gpu)
    dx, dy, mode = bm.process_batch_device(tc, tp)
    blocks = dx[0].numel()
    grp = C.c_void_p()
    _capi.check(lib.mof_shard_bm_create(C.byref(bm.cfg), None, 1, C.byref(grp)))
    try:
        slab = lib.mof_shard_bm_slab_bytes(grp, B)
        assert slab % 16 == 0 and slab >= B * (2 * blocks + 8)
        out = torch.full((slab,), -1, dtype=torch.int8, device=gpu)
        pc, pp, po = (C.c_void_p * 1)(tc.data_ptr()), (C.c_void_p * 1)(tp.data_ptr()), (C.c_void_p * 1)(out.data_ptr())
        torch.cuda.synchronize()
        _capi.check(lib.mof_shard_bm_init_gather(grp))
        _capi.check(lib.mof_shard_bm_process_batch_device(grp, pc, tc.stride(0), pp, tp.stride(0), tc.stride(1), B, po, 1))
        _capi.check(lib.mof_shard_bm_sync(grp))
        got = out.cpu().numpy()
        for k in range(B):
            ox, oy, om = C.c_size_t(), C.c_size_t(), C.c_size_t()
            _capi.check(lib.mof_shard_bm_locate(grp, B, k, C.byref(ox), C.byref(oy), C.byref(om)))
            assert np.array_equal(got[ox.value:ox.value + blocks], dx[k].cpu().numpy().ravel())
            assert np.array_equal(got[oy.value:oy.value + blocks], dy[k].cpu().numpy().ravel())
            assert np.array_equal(got[om.value:om.value + 8], mode[k].cpu().numpy().ravel())
    finally:
        lib.mof_shard_bm_destroy(grp)


def test_pair_kernel_on_the_half_tile_at_128(gpu):
    """MOF_FFT_PAIR_HALF=1: independent pairs of 128 x 128 patches through pc_pair_half_kernel (csrc/pc_seq_half.hip) -- the sequence
    kernel's passes on the half-size tile, the previous image's column spectra parked in a per-workgroup slab of device memory, two
    persistent workgroups per CU. Measured slower than the packed pair kernel (48 k against 84 k pairs/s at c4: 58 spilled VGPRs at the
    128-register limit, profiles/r05_c4_pair_half_ab.txt), so it is opt-in; a child process holds it to the oracle: 150 frame pairs of
    3 x 2 overlapping patches (more patch pairs than slabs, so every workgroup walks several), 1e-4 px on every clear-peak patch."""
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_pair_half.py")], env=dict(os.environ, MOF_FFT_PAIR_HALF="1"),
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0 and "bad 0" in r.stdout and "checked" in r.stdout, (r.stdout[-1500:], r.stderr[-1500:])


@pytest.mark.parametrize("n", [160, 150, 137, 186])
def test_half_tile_kernel_entries(gpu, n):
    """csrc/pc_half_kernel.hip is what patches of 136 .. 192 pixels run by default (r05): the batch entry on a strided view (pitch >
    width), one pair alone = the same bits as inside a batch, the video entry = the pair entry's bits. (Stateful entry, BGR8 frames, black
    frames and constant boxes on these sizes: the large-patch tests of test_gpu_generic.py / test_gpu_r04.py, which now reach this kernel.)"""
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
                tolerances.check_patch(got[k][q], want64[q], want32[q], f"half{n}/pair{k}/{kinds[k]}", q)
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
                checked += bool(tolerances.check_patch(seq[k][q], want64[q], want32[q], f"halfseq{n}/pair{k}", q))
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
                if tolerances.check_patch(seq[k][q], want64[q], want32[q], f"seqonly{n}/pair{k}", q):
                    checked += 1
    assert checked >= 0.8 * (F - 1) * gx * gy, checked

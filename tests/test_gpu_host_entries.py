"""GPU tests of the HOST-pointer batch entries (mof_fft_process_batch_host, mof_bm_process_batch_host; csrc/host_pipe.hpp): a three-slot
upload / run / download pipeline over the device batch entries -- so every form of it (pageable frames through pinned staging, pinned
frames DMA'd in place, a video uploaded once per frame, pitched rows, strided frames, ragged last chunk) must return the DEVICE entry's
bits. What these entries replace on the reference's side: a caller that holds cv::Mat frames in host memory (optic_flow.cpp:1465) and
calls processImage per frame (FftMethod.cpp:1761-1872)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SCRIPT = r"""
import os, sys
sys.path[:0] = [{root!r}]
import numpy as np, torch
from mrs_optic_flow_amd import FftMethod, BlockMethod, FastSpacedBMMethod, pinned_empty, synth
dev = torch.device("cuda", 0)
rng = np.random.default_rng(77)
checked = 0
def forms(frames):
    # every way the same n pairs can lie in host memory: (label, cur, prev)
    n = len(frames) - 1
    h, w = frames.shape[1:]
    yield "pageable pairs", frames[1:].copy(), frames[:-1].copy()
    yield "pageable video", frames[1:], frames[:-1]
    pin = pinned_empty(frames.shape); pin[:] = frames
    yield "pinned video", pin[1:], pin[:-1]
    pc, pp = pinned_empty((n, h, w)), pinned_empty((n, h, w)); pc[:] = frames[1:]; pp[:] = frames[:-1]
    yield "pinned pairs", pc, pp
    wide = np.zeros((len(frames), h, w + 13), np.uint8); wide[:, :, :w] = frames
    yield "pitched pageable video", wide[1:, :, :w], wide[:-1, :, :w]
    pw = pinned_empty(wide.shape); pw[:] = wide
    yield "pitched pinned video", pw[1:, :, :w], pw[:-1, :, :w]
    pw2 = pinned_empty((2,) + wide.shape); pw2[0] = wide; pw2[1] = wide[::-1]
    yield "pitched pinned pairs", pw2[0, 1:, :, :w], pw2[0, :-1, :, :w].copy()
    gap = pinned_empty((2 * len(frames), h, w)); gap[::2] = frames; gap[1::2] = 0
    yield "strided pinned video", gap[2::2], gap[:-2:2]
for n_frames in (18, 2, 6):
    canvas = synth.canvas_np(0, 900, 900, seed=5)
    frames = np.stack([canvas[3 * k: 3 * k + 240, 2 * k: 2 * k + 256] for k in range(n_frames)])
    frames[min(4, n_frames - 1)] = 200  # a constant frame: NaN results travel too
    fm = FftMethod(sample_point_size=64, frame_shape=(240, 256), grid=(3, 3), origin=(1, 2), stride=(60, 80))
    bm = FastSpacedBMMethod(16, 4, 4, (240, 256))
    t = torch.from_numpy(frames).to(dev)
    want = fm.process_batch_device(t[1:], t[:-1]).cpu().numpy()
    wdx, wdy, wmode = (v.cpu().numpy() for v in bm.process_batch_device(t[1:], t[:-1]))
    for label, c, p in forms(frames):
        got = fm.process_batch_host(c, p)
        assert np.array_equal(got, want, equal_nan=True), (n_frames, label)
        dx, dy, mode = bm.process_batch_host(c, p)
        assert np.array_equal(dx, wdx) and np.array_equal(dy, wdy) and np.array_equal(mode, wmode), (n_frames, label)
        checked += 1
    # a second call on the same engines (the slots are reused), pairs that are NOT a video
    perm = rng.permutation(n_frames - 1)
    c, p = np.ascontiguousarray(frames[1:][perm]), np.ascontiguousarray(frames[:-1][perm])
    assert np.array_equal(fm.process_batch_host(c, p), want[perm], equal_nan=True)
    checked += 1
# an error of the device entry inside a chunk comes back as that error, and the engine keeps working
fm = FftMethod(sample_point_size=64, frame_shape=(240, 256), grid=(3, 3), origin=(1, 2), stride=(60, 80))
try:
    fm.process_batch_host(np.zeros((3, 240, 250), np.uint8), np.zeros((3, 240, 250), np.uint8))
    raise SystemExit("a frame of the wrong shape was accepted")
except ValueError:
    pass
print("host entries ok", checked)
"""


@pytest.mark.parametrize("env", [{"MOF_HOST_CHUNK": "5"}, {"MOF_HOST_CHUNK": "1", "MOF_HOST_THREADS": "1"}, {}, {"MOF_HOST_VIDEO": "0", "MOF_HOST_CHUNK": "4"}])
def test_host_batch_entries_return_the_device_entries_bits(gpu, env):
    """Chunks of 5 (17 pairs: 5 + 5 + 5 + 2, more chunks than slots), of 1 (every slot reused many times), the default chunk (the batch
    is one chunk) and the video form switched off: all eight memory layouts x FftMethod and FastSpacedBMMethod, bit for bit."""
    script = _SCRIPT.format(root=ROOT)
    r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
    assert r.returncode == 0 and "host entries ok 27" in r.stdout, (env, r.stdout[-1500:], r.stderr[-2500:])


def test_pinned_host_memory_of_the_c_abi(gpu):
    """mof_host_alloc / register / unregister / free: a pinned array reads and writes like any other, registering a numpy buffer makes
    the library treat it as pinned (same results), and bad arguments are refused with the library's codes."""
    import ctypes as C

    import torch

    from mrs_optic_flow_amd import FftMethod, _capi, pinned_empty, synth
    lib = _capi.load()
    a = pinned_empty((7, 5), np.float64)
    a[:] = np.arange(35).reshape(7, 5)
    assert a.sum() == 35 * 34 / 2
    p = C.c_void_p()
    assert lib.mof_host_alloc(0, C.byref(p)) != 0 and lib.mof_host_register(None, 16) != 0 and lib.mof_host_free(None) == 0
    canvas = synth.canvas_np(0, 400, 400, seed=9)
    frames = np.stack([canvas[k: k + 128, 2 * k: 2 * k + 128] for k in range(9)])
    fm = FftMethod(128, 64, 80.0)
    want = fm.process_batch_host(frames[1:], frames[:-1])
    assert lib.mof_host_register(frames.ctypes.data, frames.nbytes) == 0
    try:
        got = fm.process_batch_host(frames[1:], frames[:-1])
    finally:
        assert lib.mof_host_unregister(frames.ctypes.data) == 0
    assert np.array_equal(got, want, equal_nan=True)
    t = torch.from_numpy(frames).to(gpu)
    assert np.array_equal(fm.process_batch_device(t[1:], t[:-1]).cpu().numpy(), want, equal_nan=True)


_SR_SCRIPT = r"""
import os, sys
sys.path[:0] = [{root!r}, {tests!r}]
import numpy as np, torch
import sr_scenes
from mrs_optic_flow_amd import ScaleRotationEstimator, pinned_empty
res, M, n = 240, 40.0, 23
base = sr_scenes.canvas(11, res)
video = np.stack([sr_scenes.view(base, res, 1.0 + 0.01 * t, 1.2 * t - 5.0) for t in range(n)])
video[9] = 0        # an all-zero frame: the degenerate pair and the gate (scaleRotationEstimator.cpp:119-121) are part of the contract
one = ScaleRotationEstimator(res, M)
want = np.array([one.processImage(f) for f in video])        # the frame-by-frame calls: (scale, rot)
dev = ScaleRotationEstimator(res, M)
want4 = dev.process_sequence_device(torch.from_numpy(video).cuda()).cpu().numpy()
assert np.array_equal(want4[:, :2], want)
gated0 = dev.last_gated
checked = 0
pin = pinned_empty(video.shape); pin[:] = video
wide = np.zeros((n, res, res + 9), np.uint8); wide[:, :, :res] = video
for label, v in (("pageable", video), ("pinned", pin), ("pitched", wide[:, :, :res]), ("strided", np.repeat(video, 2, 0)[::2])):
    est = ScaleRotationEstimator(res, M)
    got = est.process_sequence_host(v)
    assert np.array_equal(got, want4), label
    assert est.last_gated == gated0, (label, est.last_gated, gated0)
    # the state continues: the same video again, now on an armed estimator, in two calls == one device call of the same shape
    a = est.process_sequence_host(v[:7]); b = est.process_sequence_host(v[7:])
    again = dev.process_sequence_device(torch.from_numpy(video).cuda()).cpu().numpy() if label == "pageable" else again
    assert np.array_equal(np.concatenate([a, b]), again), label
    checked += 1
print("sr host video ok", checked)
"""


@pytest.mark.parametrize("env", [{"MOF_HOST_CHUNK": "1"}, {"MOF_HOST_CHUNK": "3", "MOF_HOST_THREADS": "2"}, {}])
def test_estimator_video_from_host_memory_is_the_frame_by_frame_calls(gpu, env):
    """mof_sr_process_sequence_host: chunks of 2 / 6 frames (a slot of the frames form holds two chunks) and the default, pageable /
    pinned / pitched / strided videos with an all-zero frame inside: the (scale, rot) of the stateful processImage loop and the (pt.x,
    pt.y) and gate count of the device video entry, bit for bit, and the state carried from one call into the next."""
    script = _SR_SCRIPT.format(root=ROOT, tests=os.path.join(ROOT, "tests"))
    r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
    assert r.returncode == 0 and "sr host video ok 4" in r.stdout, (env, r.stdout[-1500:], r.stderr[-2500:])

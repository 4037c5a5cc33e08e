"""GPU tests of the scale / rotation estimator's pipeline (csrc/sr_kernel.hip, mof_sr.hip): the log-polar remap byte for byte against
the oracle (both OpenCV generations, both interpolations), batches crossing the pipeline chunk, two streams, c5 at full size."""
import os

import numpy as np
import pytest
import torch

import oracle_lib as O
import sr_scenes
from mrs_optic_flow_amd import ScaleRotationEstimator
from mrs_optic_flow_amd.engine import INTER_CUBIC, INTER_LANCZOS4

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 1e-4
TOL = 1e-4  # px, north_star's bar for the FFT path (absolute)


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


@pytest.mark.parametrize("lanes,chunk", [(1, 0), (2, 0), (1, 1024)])
def test_c5_scale_rotation_full_size_default_passes(gpu, lanes, chunk):
    """scaleRotationEstimator.cpp:34-148 at BASELINE c5's size: 1100 pairs of 480 x 480 crops through the default
    1024-pair passes (and through 512-pair ones). Every sampled pair -- on both sides of the pass boundaries and in the ragged tail -- equals the
    same pair processed alone, bit for bit; pairs that repeat a prototype repeat its bits; six samples match the oracle
    (pt within 1e-4 px, scale / rot within 1e-5); the log-polar bytes of a sample match the oracle's byte for byte."""
    res, M, B = 480, 49.9, 1100
    base = sr_scenes.canvas(331, res)
    protos = [(1.0, 0.0), (1.03, 2.0), (0.96, -3.0), (1.0, 5.0), (1.06, -1.0), (0.92, 7.0), (1.01, 0.5), (0.99, -0.25),
              (1.02, 11.0)]
    views = np.stack([sr_scenes.view(base, res, s, r) for s, r in protos])
    P = len(protos)
    idx = np.arange(B) % P
    pidx = (idx * 4 + 1) % P
    # frames live inside 752-wide rows like the c5 crop of the camera frame (pitch 752, crop origin as the pointer)
    wide = torch.zeros((2, P, res, 752), dtype=torch.uint8, device=gpu)
    wide[:, :, :, 136:136 + res] = torch.from_numpy(views).to(gpu)
    cur = wide[0][torch.from_numpy(idx).to(gpu)][:, :, 136:136 + res]
    prev = wide[1][torch.from_numpy(pidx).to(gpu)][:, :, 136:136 + res]
    est = ScaleRotationEstimator(res, M, pipeline_lanes=lanes, batch_chunk=chunk)  # 0: the library's default pass
    got = est.process_batch_device(cur, prev)
    torch.cuda.synchronize()
    got = got.cpu().numpy()
    assert np.isfinite(got).all()
    for k in (0, 1, 255, 510, 511, 512, 513, 767, 1022, 1023, 1024, 1025, 1098, 1099):
        alone = est.process_batch_device(cur[k:k + 1], prev[k:k + 1]).cpu().numpy()[0]
        assert np.array_equal(alone, got[k]), (k, alone, got[k])
    for k in range(P, B):
        assert np.array_equal(got[k], got[k % P]), k
    for k in (0, 3, 511, 512, 1024, 1099):
        ref = O.ScaleRotationEstimator(res, M, 64)
        ref.processImage(views[pidx[k]])
        s, r = ref.processImage(views[idx[k]])
        assert np.allclose(got[k, 2:], ref.pt, rtol=0, atol=TOL), (k, got[k], ref.pt)
        assert abs(got[k, 0] - s) < 1e-5 and abs(got[k, 1] - r) < 1e-5
    # the remap stage of samples from each pass, through the same engine, byte for byte
    sample = [0, 511, 512, 1099]
    lp = est.logpolar_batch_device(cur[sample], INTER_LANCZOS4).cpu().numpy()
    lc = est.logpolar_batch_device(prev[sample], INTER_CUBIC).cpu().numpy()
    for j, k in enumerate(sample):
        assert np.array_equal(lp[j], O.logpolar(views[idx[k]], M, INTER_LANCZOS4)), k
        assert np.array_equal(lc[j], O.logpolar(views[pidx[k]], M, INTER_CUBIC)), k

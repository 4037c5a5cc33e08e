"""GPU tests of the kernel forms that are NOT the default route (A/B knobs and A/B libraries): each must pass the parity tests of
the form it replaces -- remap ring / super-tile / staged forms, the MFMA first stage of K1, the fused and split-lane estimator
column kernels, the pair kernel on the half tile at 128."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-4
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_REMAP_SCRIPT = r"""
import sys
sys.path[:0] = [{root!r}, {tests!r}]
import numpy as np, torch
import oracle_lib as O, sr_scenes
from mrs_optic_flow_amd import ScaleRotationEstimator
from mrs_optic_flow_amd.engine import INTER_CUBIC, INTER_LANCZOS4
dev = torch.device("cuda", 0)
checked = 0
for res, M in ((480, 49.9), (256, 45.0)):
    base = sr_scenes.canvas(900 + res, res)
    frames = np.stack([sr_scenes.view(base, res, 1.0 + 0.02 * k, 3.0 * k - 9.0) for k in range(9)])
    frames[4, :7, :] = 255
    est = ScaleRotationEstimator(res, M)
    t = torch.from_numpy(frames).to(dev)
    for interp in (INTER_CUBIC, INTER_LANCZOS4):
        got = est.logpolar_batch_device(t, interp).cpu().numpy()
        for k in range(9):
            want = O.logpolar(frames[k], M, interp)
            assert np.array_equal(got[k], want), (res, interp, k, int((got[k] != want).sum()))
            checked += 1
print("remap ok", checked)
"""
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("env", [{"MOF_SR_LP_RING": "16"}, {"MOF_SR_LP_SUPER": "0"}, {"MOF_SR_LP_STAGED": "0"}])
def test_remap_other_kernel_forms_are_byte_exact(gpu, env):
    """K4's 16-deep ring (maps whose largest super-tile box exceeds 3072 dwords), its one-box-per-wave form (resolutions that
    are not a multiple of 16, boxes beyond 4096 dwords) and the table-in-LDS kernel (unaligned layouts), forced by their
    knobs on maps that would take the 12-deep super-tile form: every byte against the oracle."""
    script = _REMAP_SCRIPT.format(root=ROOT, tests=os.path.join(ROOT, "tests"))
    r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
    assert r.returncode == 0 and "remap ok 36" in r.stdout, (env, r.stdout[-1500:], r.stderr[-1500:])


@pytest.mark.parametrize("env,target,select", [
    ({"MOF_SR_PAIR_SEQ": "0"}, "tests/test_gpu_sr.py", "batch_pairs or black_frames or opencv3"),
    ({"MOF_FFT_SEQ_HALF64": "1"}, "tests/test_gpu_fft_sequence.py", "video_matches or known_answers or wider_allocation"),
])
def test_knob_selected_kernel_forms_pass_their_parity_tests(gpu, env, target, select):
    """The packed pair kernels of the estimator (K5 / K6, `MOF_SR_PAIR_SEQ=0`) and the half-tile form of the 64 x 64 sequence
    kernel (`MOF_FFT_SEQ_HALF64=1`) stay in the library as A/B forms: the parity tests of the shipped forms, run once more in a
    child process with the knob set (the knobs are read once per process)."""
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, target), "-m", "gpu", "-x", "-q", "-k", select,
                        "-p", "no:cacheprovider"], capture_output=True, text=True, timeout=900, cwd=ROOT,
                       env=dict(os.environ, **env))
    assert r.returncode == 0 and " passed" in r.stdout, (env, r.stdout[-2000:], r.stderr[-1000:])


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
                          os.path.join(root, "tests", "test_gpu_sr_pipeline.py"), "-q", "-x", "-m", "gpu",
                          "-k", "not stateful and not sequence and not two_streams and not crossing_the_pipeline_chunk",  # (the fused form is an opt-in of the batch entry: a pair alone
                                                                                                                         #  runs the un-fused kernels, so batch == alone bit for bit is not its contract)
                          "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert " passed" in out.stdout


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

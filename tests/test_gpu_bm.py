"""GPU parity tests of K2/K3 (SAD block scan + histogram mode) through the C ABI. Integer work:
every comparison with the CPU oracle is bit-exact."""
import os

import numpy as np
import pytest
import torch

import oracle_lib as O
from mrs_optic_flow_amd import BlockMethod, FastSpacedBMMethod, synth

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("name", ["bm_fast_spaced_c3.npz", "bm_block_method_c1.npz"])
def test_golden_vectors(gpu, name):
    g = np.load(os.path.join(GOLDEN, name))
    block, step, radius, fast = (int(v) for v in g["params"])
    h, w = g["cur"].shape[1:]
    eng = FastSpacedBMMethod(block, radius, step, (h, w)) if fast else BlockMethod(h, block, radius)
    dx, dy, mode = eng.process_batch_host(g["cur"], g["prev"])
    assert (dx == g["dx"]).all() and (dy == g["dy"]).all()
    assert (mode[:, :2] == g["mode"]).all()
    assert (mode[:, 0:6:2] == g["top"][:, 0]).all() and (mode[:, 1:6:2] == g["top"][:, 1]).all()


@pytest.mark.parametrize("fast,shape,block,step,radius", [
    (True, (480, 752), 16, 8, 16),    # BASELINE c3
    (False, (272, 272), 32, 0, 8),    # BASELINE c1 (crop 272^2, 8x8 blocks of 32, +-8)
    (True, (100, 140), 8, 4, 5),
    (False, (120, 120), 16, 0, 21),   # default.yaml scan_radius 21
    (True, (150, 150), 64, 0, 3),     # >256 px per block: u16 partial sums are flushed
    # the reference's OWN default geometry (config/default.yaml:29-32: scan_radius 21, step_size 24, sample_point_size
    # 120) on the 752x480 camera frame and on the 480^2 crop the node hands its processors
    (True, (480, 752), 120, 24, 21),
    (True, (480, 480), 120, 24, 21),
    (False, (480, 480), 120, 0, 21),  # BlockMethod(frame_size 480, sample_point_size 120, scan_radius 21): 3 x 3 blocks
    (True, (400, 300), 128, 0, 24),   # the largest block size, at the reference's radius limit (2r + 1 <= 50, .cl:1)
    (True, (230, 420), 100, 4, 48),   # > 64 KB of LDS per workgroup: opt-in dynamic LDS
    (True, (60, 200), 8, 0, 2),       # tiny scans: eight blocks share a workgroup
    # row lengths that no chunk length of the generic scan divides (ragged row tails of 1, 3, 5 and 1 dwords) and odd radii
    (True, (200, 330), 36, 4, 7),
    (True, (180, 300), 44, 0, 9),
    (False, (250, 250), 52, 0, 11),
    (True, (330, 420), 100, 8, 13),
    (True, (300, 300), 124, 0, 5),
    (True, (90, 200), 12, 4, 3),
    (True, (90, 230), 20, 0, 6),
])
def test_seeded_batches_bit_exact(gpu, fast, shape, block, step, radius):
    h, w = shape
    B = 5
    cur, prev, shifts, kinds = synth.batch_np(B, h, w, min(radius - 2, 12), k0=0)
    if fast:
        eng, cfg = FastSpacedBMMethod(block, radius, step, shape), O.bm_config_fast_spaced(w, h, block, step, radius)
    else:
        eng, cfg = BlockMethod(h, block, radius), O.bm_config_block_method(h, block, radius)
    assert (eng.cfg.grid_x, eng.cfg.grid_y) == (cfg.grid_x, cfg.grid_y)
    dx, dy, mode = eng.process_batch_device(torch.from_numpy(cur).to(gpu), torch.from_numpy(prev).to(gpu))
    dx, dy, mode = dx.cpu().numpy(), dy.cpu().numpy(), mode.cpu().numpy()
    for k in range(B):
        wdx, wdy, wmode = O.bm_process(cur[k], prev[k], cfg)
        assert (dx[k] == wdx).all() and (dy[k] == wdy).all(), (k, kinds[k])
        assert tuple(mode[k, :2]) == wmode
        assert list(mode[k, 0:6:2]) == list(O.bm_histogram_top(wdx, radius, 3))
        assert list(mode[k, 1:6:2]) == list(O.bm_histogram_top(wdy, radius, 3))
        if kinds[k] == "shift":
            assert wmode == (-shifts[k][0], -shifts[k][1])  # block matching reports the opposite sign


def test_ties_and_low_contrast_rule(gpu):
    f = np.full((112, 112), 90, np.uint8)
    dx, dy, mode = BlockMethod(112, 32, 8).process_batch_host(f[None], f[None])
    assert (dx == -8).all() and (dy == -8).all() and tuple(mode[0, :2]) == (-8, -8)  # BlockMethod.cpp:63 first min
    dx, dy, mode = FastSpacedBMMethod(16, 8, 8, (112, 112)).process_batch_host(f[None], f[None])
    assert (dx == 0).all() and (dy == 0).all()                                      # FastSpacedBMMethod.cl:77-82
    rng = np.random.default_rng(7)
    for _ in range(8):  # tiny alphabet -> many exact ties; the first minimum in row-major order must win
        prev = rng.integers(0, 3, (40, 40), dtype=np.uint8)
        cur = rng.integers(0, 3, (40, 40), dtype=np.uint8)
        cfg = O.bm_config_block_method(40, 8, 4)
        dx, dy, _ = BlockMethod(40, 8, 4).process_batch_host(cur[None], prev[None])
        wdx, wdy, _ = O.bm_process(cur, prev, cfg)
        assert (dx[0] == wdx).all() and (dy[0] == wdy).all()
    r, b = 5, 8  # threshold 5.0 exactly: gap 5 zeroed, gap 6 kept (double compare, .cl:2)
    size = b + 2 * r
    for gap in (5, 6):
        prev = np.full((size, size), 100, np.uint8)
        prev[r, r] = 100 + gap
        cur = np.full((size, size), 100, np.uint8)
        dx, dy, _ = FastSpacedBMMethod(b, r, 0, (size, size)).process_batch_host(cur[None], prev[None])
        wdx, wdy, _ = O.bm_process(cur, prev, O.bm_config_fast_spaced(size, size, b, 0, r))
        assert (dx[0] == wdx).all() and (dy[0] == wdy).all()


def test_stateful_processimage(gpu):
    """prev starts as zeros (BlockMethod.cpp:17-18), then prev <- cur after every call (:89)."""
    fs = 144
    seq = [synth.pair_np(5, fs, fs, 2 * t, -t, blur=False)[0] for t in range(3)]
    eng = BlockMethod(fs, 32, 8)
    cfg = O.bm_config_block_method(fs, 32, 8)
    prev = np.zeros((fs, fs), np.uint8)
    for f in seq:
        dx, dy, mode = eng.processBlocks(f)
        wdx, wdy, wmode = O.bm_process(f, prev, cfg)
        assert (dx == wdx).all() and (dy == wdy).all() and mode == wmode
        prev = f
    assert eng.processImage(seq[0]).shape == (1, 2)


def test_full_size_c3_batch_properties(gpu):
    """BASELINE c3 at full size (752x480, sps 16, step 8, r 16, batch 1024)."""
    B, h, w = 1024, 480, 752
    cur, prev, shifts, kinds = synth.batch_torch(B, h, w, 12, gpu)
    eng = FastSpacedBMMethod(16, 16, 8, (h, w))
    dx, dy, mode = eng.process_batch_device(cur, prev)
    torch.cuda.synchronize()
    dx, dy, mode, sh = dx.cpu().numpy(), dy.cpu().numpy(), mode.cpu().numpy(), shifts.numpy()
    for k in range(B):
        if kinds[k] == "shift":
            assert (dx[k] == -sh[k, 0]).all() and (dy[k] == -sh[k, 1]).all()  # SAD 0 at the planted offset
            assert tuple(mode[k, :2]) == (-sh[k, 0], -sh[k, 1])
        elif kinds[k] in ("identical", "constant"):
            assert (dx[k] == 0).all() and (dy[k] == 0).all()
    cfg = O.bm_config_fast_spaced(w, h, 16, 8, 16)
    for k in (11, 531):  # noisy pairs against the oracle
        wdx, wdy, wmode = O.bm_process(cur[k].cpu().numpy(), prev[k].cpu().numpy(), cfg)
        assert (dx[k] == wdx).all() and (dy[k] == wdy).all() and tuple(mode[k, :2]) == wmode


def test_random_geometries_bit_exact(gpu):
    """Seeded sweep over block sizes / steps / radii / frame sizes (fast 16x16 path and the generic kernel)."""
    rng = np.random.default_rng(20261003)
    for trial in range(24):
        block = int(rng.choice([4, 8, 12, 16, 16, 16, 20, 32, 72, 120]))
        radius = int(rng.choice([2, 5, 8, 8, 16, 16, 21])) if block != 16 else int(rng.choice([8, 16, 16, 5]))
        step = int(rng.choice([0, 4, 8, 3])) if block != 16 else int(rng.choice([0, 4, 8, 12]))
        gx, gy = int(rng.integers(1, 9)), int(rng.integers(1, 5))
        S = block + step
        w = gx * S + 2 * radius + int(rng.integers(0, S))  # reference grid maths: (w - 2r) / S = gx
        h = gy * S + 2 * radius + int(rng.integers(0, S))
        cur = rng.integers(0, 256, (2, h, w), dtype=np.uint8)
        prev = np.roll(cur, (int(rng.integers(-3, 4)), int(rng.integers(-3, 4))), axis=(1, 2))
        prev = np.clip(prev.astype(np.int32) + rng.integers(-6, 7, prev.shape), 0, 255).astype(np.uint8)
        eng = FastSpacedBMMethod(block, radius, step, (h, w))
        cfg = O.bm_config_fast_spaced(w, h, block, step, radius)
        assert (eng.cfg.grid_x, eng.cfg.grid_y) == (cfg.grid_x, cfg.grid_y)
        dx, dy, mode = eng.process_batch_host(cur, prev)
        for k in range(2):
            wdx, wdy, wmode = O.bm_process(cur[k], prev[k], cfg)
            assert (dx[k] == wdx).all() and (dy[k] == wdy).all(), (trial, block, step, radius, w, h)
            assert tuple(mode[k, :2]) == wmode


@pytest.mark.parametrize("block,step,radius", [(16, 8, 16), (16, 4, 8), (24, 4, 5), (120, 24, 21), (32, 0, 8), (12, 4, 3)])
def test_bgr_front_end_fused_into_the_block_scans(gpu, block, step, radius):
    """SURVEY N2 for K2: interleaved BGR8 camera frames (a crop of a larger frame, odd byte offsets), CV_RGB2GRAY as the node
    applies it (optic_flow.cpp:1622) inside the staging loads of both scan kernels: the same bits as the gray entry on the
    converted crop, and the oracle's shifts."""
    B, h, w, xi, yi = 3, 150 + 2 * radius, 260 + 2 * radius, 7, 5
    if block == 120:
        h, w = 330, 480
    H, W = h + 11, w + 19
    rng = np.random.default_rng(block * 100 + radius)
    base = [synth.pair_np(30 + k, H, W, 2 + k, -k) for k in range(B)]

    def colour(g):
        g = g.astype(np.int32)
        ch = np.stack([g, 255 - g // 2, (g * 3 // 4 + 20)], axis=-1) + rng.integers(-2, 3, g.shape + (3,))
        return np.clip(ch, 0, 255).astype(np.uint8)
    cur = np.stack([colour(c) for c, _ in base])
    prev = np.stack([colour(p) for _, p in base])
    eng = FastSpacedBMMethod(block, radius, step, (h, w))
    cfg = O.bm_config_fast_spaced(w, h, block, step, radius)
    tc, tp = torch.from_numpy(cur).to(gpu), torch.from_numpy(prev).to(gpu)
    dx, dy, mode = eng.process_batch_device_bgr(tc[:, yi:yi + h, xi:xi + w], tp[:, yi:yi + h, xi:xi + w])
    dx, dy, mode = dx.cpu().numpy(), dy.cpu().numpy(), mode.cpu().numpy()
    for k in range(B):
        gc, gp = O.rgb2gray(cur[k, yi:yi + h, xi:xi + w]), O.rgb2gray(prev[k, yi:yi + h, xi:xi + w])
        wdx, wdy, wmode = O.bm_process(gc, gp, cfg)
        assert (dx[k] == wdx).all() and (dy[k] == wdy).all() and tuple(mode[k, :2]) == wmode, k
        gx, gy, gm = eng.process_batch_device(torch.from_numpy(gc[None]).to(gpu), torch.from_numpy(gp[None]).to(gpu))
        assert torch.equal(gx[0].cpu(), torch.from_numpy(dx[k])) and torch.equal(gy[0].cpu(), torch.from_numpy(dy[k]))
        assert np.array_equal(gm[0].cpu().numpy(), mode[k])


def test_refine_matches_oracle(gpu):
    """mof_bm_refine (BlockMethod::Refine, faithful and repaired) on the frames of the last processImage call."""
    fs = 144
    seq = [synth.pair_np(81, fs, fs, -3 * t, 2 * t)[0] for t in range(3)]
    eng = BlockMethod(fs, 32, 8)
    eng.processBlocks(seq[0])
    for t in (1, 2):
        dx, dy, mode = eng.processBlocks(seq[t])
        for faithful in (True, False):
            for fp in (mode, (0, 0), (-2, -1), (3, -4)):
                want = O.bm_refine(seq[t], seq[t - 1], fp, 2, faithful)
                assert eng.refine(fp, 2, faithful) == want, (t, faithful, fp)
    from mrs_optic_flow_amd import MofError
    with pytest.raises(MofError):
        eng.refine((400, 0))


@pytest.mark.parametrize("radius", [2, 4, 6, 8, 10, 12, 14, 16])
def test_fast_16x16_scan_at_every_even_radius(gpu, radius):
    """r06: bm_scan16_kernel<R> (blocks of 16 x 16, v_qsad_pk_u16_u8 on register-resident blocks) serves every even scan radius up to 16,
    not only c3's 8 and 16. Bit-exact against the oracle on random frames with ties (small alphabet), the low-contrast rule on, steps 0 / 4 / 8,
    block rows that fill a wave exactly, spill into a second one, or hold a single block; gray and BGR8."""
    rng = np.random.default_rng(100 + radius)
    for step, gx, gy, extra in ((0, 1, 1, 0), (4, 5, 2, 3), (8, 64 // max(1, radius // 2) + 1, 1, 7), (8, 3, 3, 0), (0, 9, 2, 11)):
        S = 16 + step
        w, h = gx * S + 2 * radius + extra % S, gy * S + 2 * radius + (extra % 5)  # (FastSpacedBMMethod_OCL.cpp:82-90: (size - 2 r) / S blocks per axis)
        levels = 256 if step else 4  # (a four-level alphabet: many exact ties, the first minimum in row-major order must win)
        cur = rng.integers(0, levels, (3, h, w), dtype=np.uint8)
        prev = np.roll(cur, (int(rng.integers(-2, 3)), int(rng.integers(-2, 3))), axis=(1, 2))
        prev = np.clip(prev.astype(np.int32) + rng.integers(-1, 2, prev.shape), 0, 255).astype(np.uint8)
        eng = FastSpacedBMMethod(16, radius, step, (h, w))
        assert (eng.cfg.grid_x, eng.cfg.grid_y) == (gx, gy), (eng.cfg.grid_x, eng.cfg.grid_y, gx, gy)
        cfg = O.bm_config_fast_spaced(w, h, 16, step, radius)
        dx, dy, mode = (v.cpu().numpy() for v in eng.process_batch_device(torch.from_numpy(cur).to(gpu), torch.from_numpy(prev).to(gpu)))
        for k in range(3):
            wdx, wdy, wmode = O.bm_process(cur[k], prev[k], cfg)
            assert (dx[k] == wdx).all() and (dy[k] == wdy).all() and tuple(mode[k, :2]) == wmode, (radius, step, gx, gy, k)

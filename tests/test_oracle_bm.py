"""CPU tests of the block-matching oracle (oracle/bm_ref.c): twin, known answers, tie-breaking,
the low-contrast rule, histogram order, golden vectors. Integer work: every comparison is exact."""
import os

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

import oracle_lib as O
import twin
from mrs_optic_flow_amd import synth

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def test_geometry_helpers_follow_the_reference():
    c = O.bm_config_block_method(272, 32, 8)  # BlockMethod.cpp:11 -> (272-16)/32 = 8
    assert (c.grid_x, c.grid_y, c.step, c.low_contrast_rule) == (8, 8, 0, 0)
    c = O.bm_config_fast_spaced(752, 480, 16, 8, 16)  # FastSpacedBMMethod_OCL.cpp:90 -> 30 x 18
    assert (c.grid_x, c.grid_y, c.low_contrast_rule) == (30, 18, 1)


@pytest.mark.parametrize("fast", [False, True])
def test_matches_numpy_twin(fast):
    h, w = (120, 168) if fast else (112, 112)
    for k in range(4):
        dx0, dy0 = synth.planted_shift(k + 1, 6)
        cur, prev = synth.pair_np(k, h, w, dx0, dy0, kind="noisy" if k == 3 else "shift")
        cfg = O.bm_config_fast_spaced(w, h, 16, 8, 10) if fast else O.bm_config_block_method(h, 32, 8)
        dx, dy, mode, sad = O.bm_process(cur, prev, cfg, want_sad=True)
        tdx, tdy, tmode, tsad = twin.bm_process(cur, prev, cfg.block, cfg.step, cfg.radius, (cfg.grid_x, cfg.grid_y),
                                                bool(cfg.low_contrast_rule))
        assert (sad.reshape(tsad.shape) == tsad).all()
        assert (dx == tdx).all() and (dy == tdy).all() and mode == tmode


def test_integer_translation_is_found_exactly_with_opposite_sign():
    # cur(y,x) = prev(y-dy, x-dx): the cur block matches prev displaced by (-dx,-dy) with SAD 0
    h = w = 144
    cur, prev = synth.pair_np(7, h, w, 5, -3, blur=False)
    cfg = O.bm_config_block_method(h, 32, 8)
    dx, dy, mode, sad = O.bm_process(cur, prev, cfg, want_sad=True)
    assert (dx == -5).all() and (dy == 3).all() and mode == (-5, 3)
    assert (sad.reshape(cfg.grid_y * cfg.grid_x, -1).min(axis=1) == 0).all()


def test_constant_frames_tie_breaking():
    f = np.full((112, 112), 90, np.uint8)
    # BlockMethod: every SAD is 0 -> first minimum = top-left of the scan = (-r,-r)  (BlockMethod.cpp:63)
    dx, dy, mode = O.bm_process(f, f, O.bm_config_block_method(112, 32, 8))
    assert (dx == -8).all() and (dy == -8).all() and mode == (-8, -8)
    # FastSpacedBM: SAD(0,0) - min = 0 <= 0.2 r^2 -> (0,0)  (FastSpacedBMMethod.cl:77-82)
    dx, dy, mode = O.bm_process(f, f, O.bm_config_fast_spaced(112, 112, 16, 8, 8))
    assert (dx == 0).all() and (dy == 0).all() and mode == (0, 0)


def test_low_contrast_rule_boundary():
    """r = 5 -> threshold 25*0.2 = 5.0: a centre SAD exactly 5 above the minimum is still zeroed, 6 is not."""
    r, b = 5, 8
    size = b + 2 * r
    for gap, zeroed in ((5, True), (6, False)):
        prev = np.full((size, size), 100, np.uint8)
        cur = np.full((size, size), 100, np.uint8)
        # one pixel of the window seen only by candidate (xs,ys) = (r,r) [the centre], value raised by `gap`
        prev[r, r] = 100 + gap  # candidate (r,r) reads prev[r + j, r + i]; pixel (i=0,j=0)
        # every candidate whose footprint covers prev[r, r] pays `gap`; candidates with xs > r or ys > r do not
        cfg = O.bm_config_fast_spaced(size, size, b, 0, r)
        dx, dy, _, sad = O.bm_process(cur, prev, cfg, want_sad=True)
        assert sad[0, r, r] - sad[0].min() == gap
        assert ((dx[0, 0], dy[0, 0]) == (0, 0)) == zeroed


@settings(max_examples=25, deadline=None)
@given(st.integers(0, 2**31 - 1))
def test_argmin_is_first_in_row_major_order(seed):
    rng = np.random.default_rng(seed)
    r, b = 3, 4
    size = b + 2 * r
    prev = rng.integers(0, 3, (size, size), dtype=np.uint8)  # tiny alphabet -> many ties
    cur = rng.integers(0, 3, (size, size), dtype=np.uint8)
    cfg = O.bm_config_block_method(size, b, r)
    dx, dy, _, sad = O.bm_process(cur, prev, cfg, want_sad=True)
    k = int(np.argmin(sad[0]))  # numpy's argmin is first-occurrence, row-major
    assert (dy[0, 0] + r, dx[0, 0] + r) == divmod(k, 2 * r + 1)


def test_histogram_top_is_stable_descending():
    d = np.array([2, 2, -1, -1, 0, 3, 3, -3], np.int8)  # counts: 2:2, -1:2, 3:2, 0:1, -3:1
    assert list(O.bm_histogram_top(d, 3, 3)) == [-1, 2, 3]  # ties keep the smaller shift first (.cl:126-151)
    assert list(O.bm_histogram_top(np.array([1, 1, 1], np.int8), 2, 3)) == [1, -2, -1]


def test_golden_vectors_reproduce():
    files = sorted(f for f in os.listdir(GOLDEN) if f.startswith("bm_") and f.endswith(".npz"))
    assert files
    for f in files:
        g = np.load(os.path.join(GOLDEN, f))
        block, step, radius, fast = (int(v) for v in g["params"])
        h, w = g["cur"].shape[1:]
        cfg = O.bm_config_fast_spaced(w, h, block, step, radius) if fast else O.bm_config_block_method(h, block, radius)
        for k in range(g["cur"].shape[0]):
            dx, dy, mode = O.bm_process(g["cur"][k], g["prev"][k], cfg)
            assert (dx == g["dx"][k]).all() and (dy == g["dy"][k]).all() and mode == tuple(g["mode"][k])


def test_resize_2x_fixed_point():
    """cv::resize x2 (INTER_LINEAR, CV_8UC1): interior = (1/4, 3/4) taps in both axes, borders replicate."""
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (7, 9), dtype=np.uint8)
    up = O.resize_2x(img)
    assert up.shape == (14, 18)
    f = img.astype(np.float64)
    # interior destination (2k, 2l): 1/16 * (a(k-1,l-1) + 3 a(k-1,l) + 3 a(k,l-1) + 9 a(k,l)), within fixed-point rounding
    for k, l in [(2, 3), (4, 5), (1, 1)]:
        ref = (f[k - 1, l - 1] + 3 * f[k - 1, l] + 3 * f[k, l - 1] + 9 * f[k, l]) / 16
        assert abs(float(up[2 * k, 2 * l]) - ref) <= 1.0
    assert up[0, 0] == img[0, 0] and up[-1, -1] == img[-1, -1]        # corners replicate
    assert (O.resize_2x(np.full((5, 5), 200, np.uint8)) == 200).all()  # weights sum to one


def test_refine_follows_the_reference_text():
    """BlockMethod::Refine, literally (BlockMethod.cpp:96-147): with the F9 defect both 2x images are the current
    frame, so a non-negative offset is returned unchanged while a negative one drifts."""
    cur, prev = synth.pair_np(13, 96, 96, 3, -2)
    assert O.bm_refine(cur, prev, (0, 0), 2, True) == (0.0, 0.0)
    assert O.bm_refine(cur, prev, (2, 3), 2, True) == (2.0, 3.0)
    (x, y), sads = O.bm_refine(cur, prev, (-3, 2), 2, True, want_sads=True)
    # a negative offset shifts the cut-out of the (self-)"previous" image: every candidate compares the image with a
    # shifted copy of itself and the offset drifts (known answer of this restatement)
    assert (x, y) == (-3.25, 1.75) and sads.shape == (2, 3, 3) and sads[0, 1, 1] > 0
    # repaired variant: previous image really is the previous frame; identical frames refine to exactly zero offset
    assert O.bm_refine(cur, cur, (0, 0), 2, False) == (0.0, 0.0)
    with pytest.raises(ValueError):
        O.bm_refine(cur, prev, (200, 0), 2, True)  # cut-out would be empty

"""The ONE statement of the FFT path's parity bars (BASELINE north_star: sub-pixel shifts within 1e-4 px of the reference's CPU path).

Two oracles exist: the f64 restatement ("truth") and the f32 one -- the reference's own arithmetic (cv::phaseCorrelate computes in
CV_32F, /root/reference/src/FftMethod.cpp:1805-1806, :1836). On almost every patch they agree to ~2e-7 px and a kernel is held to
1e-4 px against BOTH (the fast path below: nothing else is computed).

A patch that misses that -- the oracles more than 2e-5 px apart, or the kernel more than 1e-4 px from either -- is classified from
its INPUT PIXELS alone (tests/conditioning.py): bins that are zero in exact arithmetic, bins under the f32 rounding floor, how far the
5 x 5 centroid's denominator cancels, and -- the measurement that needs no model -- `spread`: how far five independent f32 transform
libraries / orders (pocketfft complex64, the same on the transposed patch, pocketfft's real transform, torch.fft complex and real) land
from the f64 pipeline on that very patch. The C oracle's f32 variant is a sixth such implementation (`dd` = its distance from the f64
oracle). The rule:

    zero_term = (exact-zero bins, when there are 1..16 of them) x 2 sqrt 2 / |window sum|   (what a transform that does not cancel them adds,
                                                                                              even if every library here happens to)
    allowance = max(2 x max(spread, dd), zero_term)
    bar       = 1e-4 + allowance,  never above 1e-3 px;  asserted against the f32 oracle AND the f64 oracle
    unpinned  = 1e-4 + max(2 x spread, zero_term)  >  1e-3 px      (computed WITHOUT dd and without the kernel)

i.e. a kernel may be as far from either oracle as correct f32 transforms demonstrably are from the truth on that input (x 2: six samples
under-estimate a tail), and no further than 1e-3 px; where independent f32 libraries themselves scatter beyond that, the reference's
arithmetic pins nothing (which f32 answer OpenCV's radix order gives cannot be known here: OpenCV is absent) and only the integer peak
is asserted. One more case exists: the two ORACLES themselves further apart than the ceiling allows (1e-4 + 2 dd > 1e-3) on a patch whose
libraries do not scatter that far -- then no answer satisfies both, and the kernel is held to EITHER restatement at the bar of its inputs
(rule "oracles apart: held to either"; it never occurred in the test suite, the large-band fuzz met it twice). What the table of records shows (profiles/r06_f32_limited.json): the exact-zero-bin patches split the libraries -- on
fs480/n48 #89 pocketfft cancels both bins and lands 2e-8 px from f64 while torch.fft and the f32 oracle land 1e-3 / 8e-4 px away, on
fs480/n60 #22 pocketfft 9e-7, torch 3e-4 .. 7e-4 -- so "any f32 order is that far off" (the r05 text) was wrong: SOME orders are exact there and
some are not, and that is precisely why the patch is unpinned. Every patch that leaves the fast path is recorded with all columns;
conftest.py writes the session's JSON and bounds the counts (every module's records)."""
import numpy as np

TOL = 1e-4                 # px, against both oracles
F32_LIMITED_FROM = 2e-5    # px of oracle-to-oracle distance above which a patch leaves the fast path even if the kernel is within TOL
SPREAD_FACTOR = 2.0
CEILING = 1e-3             # px: no asserted sub-pixel bar ever exceeds this
ZERO_BINS_FEW = 16         # the per-bin allowance applies to a FEW coincidental zeros; whole zero rows (constant / checker patches) are
                           # cancelled exactly by every transform or show in `spread`
MAX_UNPINNED = 1           # unpinned patches a session may meet OUTSIDE the seeded one-per-mechanism tests (label prefix "mechanism/")
MAX_RELAXED = 40           # patches off the fast path per session (of ~60,000 checked)
RECORDS = []               # dicts, one per patch that left the fast path; dumped and bounded by conftest.py


def allowance(info, dd):
    few = info["zero_bin_px"] if 0 < info["zero_bins"] <= ZERO_BINS_FEW else 0.0
    return max(SPREAD_FACTOR * max(info["spread_px"], dd), few), TOL + max(SPREAD_FACTOR * info["spread_px"], few) > CEILING


def check_patch(got, want64, want32, label, patch, pixels=None, what="kernel"):
    """Assert one patch's (x, y) against the bars above. `pixels` = (cur patch, prev patch) uint8 or a callable returning them (only
    evaluated off the fast path). NaN patterns must agree wherever the two oracles agree on them. Returns True if the patch was
    pinned (compared at a sub-pixel bar), False if it is unpinned or the oracles disagree about validity."""
    got, want64, want32 = np.asarray(got, np.float64), np.asarray(want64, np.float64), np.asarray(want32, np.float64)
    n64, n32 = np.isnan(want64), np.isnan(want32)
    if n64.any() or n32.any():
        if np.array_equal(n64, n32):
            assert np.array_equal(np.isnan(got), n64), (label, patch, got, want64)
            return True
        return False
    assert not np.isnan(got).any(), (label, patch, what, got, want64)
    dd = float(np.abs(want32 - want64).max())
    e64, e32 = float(np.abs(got - want64).max()), float(np.abs(got - want32).max())
    if dd <= F32_LIMITED_FROM and e64 <= TOL and e32 <= TOL:
        return True
    assert pixels is not None, (label, patch, what, "off the fast path and no pixels to classify", got, want64, want32)
    import conditioning

    a, b = pixels() if callable(pixels) else pixels
    info = conditioning.analyse(a, b)
    allow, unpinned = allowance(info, dd)
    bar = None if unpinned else min(TOL + allow, CEILING)
    RECORDS.append({"label": label, "patch": int(patch), "what": what, "rule": "unpinned by inputs" if unpinned else "relaxed by inputs",
                    "mechanism": info["mechanism"], "transform_size": info["transform_size"], "zero_bins": info["zero_bins"],
                    "zero_bin_list": info["zero_bin_list"], "floor_bins": info["floor_bins"], "cancellation": info["cancellation"],
                    "peak_over_m2": info["peak_over_m2"], "independent_f32_minus_f64_px": info["independent_f32_minus_f64_px"],
                    "spread_px": info["spread_px"], "oracle_distance_px": dd, "bar_px": bar,
                    "kernel_minus_f32_oracle_px": e32, "kernel_minus_f64_oracle_px": e64})
    if unpinned:
        assert e32 <= 0.25 and e64 <= 0.25, (label, patch, what, got, want32, want64, info)  # the integer peak still agrees
        return False
    if TOL + SPREAD_FACTOR * dd > CEILING:
        # The two ORACLES are further apart than the ceiling lets a kernel be from both (found by the large-band fuzz, seed 901: a constant
        # frame against texture at 363 -> 375; f32 oracle 5.3e-3 px from the f64 one, kernel 2e-6 px from the f64 one, the libraries of
        # that box within 4.5e-4): no answer can satisfy both. The kernel must agree with ONE of the two restatements at the bar its
        # inputs give, and with both on the integer peak.
        bar_in = min(TOL + max(SPREAD_FACTOR * info["spread_px"], info["zero_bin_px"] if 0 < info["zero_bins"] <= ZERO_BINS_FEW else 0.0), CEILING)
        RECORDS[-1]["rule"] = "oracles apart: held to either"
        RECORDS[-1]["bar_px"] = bar_in
        assert min(e32, e64) <= bar_in and e32 <= 0.25 and e64 <= 0.25, (label, patch, what, got, want32, want64, dd, bar_in, info)
        return True
    assert e32 <= bar and e64 <= bar, (label, patch, what, got, want32, want64, dd, bar, info)
    return True


def check_patch_strict(got, want64, want32, label, patch, what="kernel"):
    """The STRICTER rule a test may choose where the kernel is built to give the exact-arithmetic answer although f32 libraries
    scatter (a constant frame against texture on a padded size: the kernel takes the constant box from its closed form and zeroes the
    box's exact-zero lines, as the f64 oracle effectively does): against the f32 oracle at 1e-4 + 2 x the oracle-to-oracle distance, never
    above 1e-3 px, no unpinned escape. Recorded when the relaxation is used."""
    got, want64, want32 = np.asarray(got, np.float64), np.asarray(want64, np.float64), np.asarray(want32, np.float64)
    assert not (np.isnan(want64).any() or np.isnan(want32).any() or np.isnan(got).any()), (label, patch, got, want64, want32)
    dd = float(np.abs(want32 - want64).max())
    e64, e32 = float(np.abs(got - want64).max()), float(np.abs(got - want32).max())
    if dd <= F32_LIMITED_FROM:
        assert e64 <= TOL and e32 <= TOL, (label, patch, what, got, want64, want32)
        return True
    bar = min(TOL + SPREAD_FACTOR * dd, CEILING)
    RECORDS.append({"label": label, "patch": int(patch), "what": what, "rule": "strict: oracle distance only", "mechanism": "(not classified: the test demands the exact-arithmetic answer)",
                    "oracle_distance_px": dd, "bar_px": bar, "kernel_minus_f32_oracle_px": e32, "kernel_minus_f64_oracle_px": e64})
    assert e32 <= bar, (label, patch, what, got, want32, want64, dd, bar)
    return True


def patch_pixels(cur, prev, lay, p):
    """(cur patch, prev patch) of patch index p = i + j * grid_x under an oracle_lib.FftLayout, as a thunk for check_patch."""
    n, gx = lay.patch, lay.grid_x
    x0, y0 = lay.origin_x + (p % gx) * lay.stride_x, lay.origin_y + (p // gx) * lay.stride_y
    return lambda: (cur[y0:y0 + n, x0:x0 + n], prev[y0:y0 + n, x0:x0 + n])


def check_frame(got, cur, prev, lay, label, what="kernel", only_stable=True, oracles=None):
    """One frame pair's [gy * gx, 2] result against both oracles, patch by patch. Patches whose correlation surface has no clear peak
    (second-highest value outside the 5 x 5 window >= half the peak: the arg-max is decided by rounding noise) are skipped. Returns the
    number of patches pinned."""
    import oracle_lib as O

    if oracles is None:
        want64, _, diags = O.fft_process(cur, prev, lay, 64, want_diag=True)
        want32, _ = O.fft_process(cur, prev, lay, 32)
    else:
        want64, want32, diags = oracles
    pinned = 0
    for p in range(want64.shape[0]):
        if only_stable and not diags[p].second_value < 0.5 * diags[p].peak_value:
            continue
        pinned += bool(check_patch(got[p], want64[p], want32[p], label, p, what=what, pixels=patch_pixels(cur, prev, lay, p)))
    return pinned


def summary():
    """Counts for the session's JSON and the bounds conftest.py / test_zz_session_records.py assert."""
    outside = [r for r in RECORDS if not r["label"].startswith("mechanism/")]
    unpinned_outside = {(r["label"].rsplit("/", 1)[0] if r["label"].count("/") else r["label"], r["patch"]) for r in outside if r["bar_px"] is None}
    return {"tol_px": TOL, "spread_factor": SPREAD_FACTOR, "ceiling_px": CEILING, "count": len(RECORDS),
            "relaxed_outside_mechanism_tests": sum(r["bar_px"] is not None for r in outside),
            "unpinned_outside_mechanism_tests": len(unpinned_outside),
            "unpinned": sum(r["bar_px"] is None for r in RECORDS),
            "worst_bar_px": max([r["bar_px"] for r in RECORDS if r["bar_px"] is not None], default=None),
            "by_mechanism": {m: sum(r["mechanism"] == m for r in RECORDS) for m in sorted({r["mechanism"] for r in RECORDS})}}


def violations():
    s = summary()
    bad = []
    if s["unpinned_outside_mechanism_tests"] > MAX_UNPINNED:
        bad.append(f"{s['unpinned_outside_mechanism_tests']} unpinned patches outside the mechanism tests (allowed {MAX_UNPINNED})")
    if s["relaxed_outside_mechanism_tests"] > MAX_RELAXED:
        bad.append(f"{s['relaxed_outside_mechanism_tests']} relaxed patches (allowed {MAX_RELAXED})")
    if any(r["bar_px"] is not None and r["bar_px"] > CEILING for r in RECORDS):
        bad.append("a bar above the ceiling")
    return bad

import sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import oracle_lib as O
from mrs_optic_flow_amd import FftMethod, synth
for fs, n in [(480, 60), (480, 48), (480, 30), (480, 80)]:
    fm = FftMethod(fs, n, 80.0)
    sq = fs // n
    seq = [synth.pair_np(40 + n, fs, fs, 2 * t, -t, blur=True)[0] for t in range(3)]
    lay = O.fft_layout(fs, fs, n, sq, sq)
    fm.processImage(seq[0])
    for t in (1, 2):
        out = fm.processImage(seq[t])
        w64, _, diags = O.fft_process(seq[t], seq[t - 1], lay, 64, want_diag=True)
        w32, _ = O.fft_process(seq[t], seq[t - 1], lay, 32)
        well = np.array([d.second_value < 0.5 * d.peak_value for d in diags])
        d64 = np.abs(out - w64).max(axis=1); d32 = np.abs(out - w32).max(axis=1); dd = np.abs(w32 - w64).max(axis=1)
        bad = np.where(well & (d64 > 1e-4))[0]
        print(fs, n, t, "well", well.sum(), "/", well.size, "max d64", np.nanmax(d64[well]), "max d32", np.nanmax(d32[well]), "max 32v64", np.nanmax(dd[well]))
        for p in bad[:6]:
            print("   patch", p, out[p], w64[p], w32[p], "peak", diags[p].peak_value, "second", diags[p].second_value)

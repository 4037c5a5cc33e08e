"""LDS bank-conflict model of K1 at N = 128 with FREE lane maps (r03): which of a wave's lines / butterfly indices share a
32-lane read group or a 16-lane write group is a choice that costs no arithmetic as long as the twiddle index x stays a
function of the lane alone. Model as lds_conflicts_120.py (ds_read_b64: 32-lane groups over 64 dword banks; ds_write_b64:
16-lane groups over 32 banks; 8-byte elements).

  python lds_conflicts_128.py            current layout (pitch 136) and the r03 layout (pitch 140, 8 (r >> 5) row skew,
                                         class-ordered lines) side by side
Derivation (DESIGN.md, K1 N = 128): with pitch = 12 (mod 32) the rows' first banks 12 r mod 32 run through all eight
multiples of 4; even rows are the multiples of 8, odd rows the rest, and both classes are closed under r -> -r (mod 8),
which the Hermitian reads of the inverse column pass need.
"""
import sys
import numpy as np
from lds_conflicts_120 import cost

N, H, R1, R2, WAVES, LPW, LI, WI = 128, 64, 16, 8, 16, 8, 8, 8
ORDER_S1 = np.array([0, 2, 4, 6, 1, 3, 5, 7])  # 32-lane groups = one class; 16-lane pairs 8 apart mod 16
ORDER_S2 = np.array([0, 4, 2, 6, 1, 5, 3, 7])  # 32-lane groups = two lines 16 apart mod 32


def evaluate(new):
    if new:
        za = lambda r, c: r * 140 + 8 * (r >> 5) + c + (c >> 4)
        o1, o2 = ORDER_S1, ORDER_S2
    else:
        za = lambda r, c: r * 136 + c + (c >> 4)
        o1 = o2 = np.arange(8)
    lane = np.arange(64)
    allon = np.ones(64, bool)
    tot = {}

    def add(name, addrs, active, write):
        t, i = cost(np.asarray(addrs), active, write)
        a = tot.setdefault(name, [0, 0])
        a[0] += t
        a[1] += i

    def row_pass(tag, line0):
        line, x = line0 + o1[lane // R2], lane % R2
        for k in range(R1):
            add(tag + " s1 r", za(line, x + k * R2), allon, False)
            add(tag + " s1 w", za(line, x * R1 + k), allon, True)
        for b in range(2):
            q = lane + 64 * b
            line, x = line0 + o2[q // R1], q % R1
            for k in range(R2):
                add(tag + " s2 r", za(line, x + k * R1), allon, False)
                add(tag + " s2 w", za(line, x + k * R1), allon, True)

    for wave in range(WAVES):
        row, col = wave * LPW + o1[lane // 8], (lane % 8) * 16
        for i in range(16):
            add("load w", za(row, col + i), allon, True)
        row_pass("row", wave * LPW)
        col0 = wave * LPW
        col, x = col0 + lane % 8, o1[lane // 8]
        for k in range(R1):
            add("col s1 r", za(x + k * R2, col), allon, False)
            add("col s1 w", za(x * R1 + k, col), allon, True)
        for b in range(2):
            col, x = col0 + lane % 4 + 4 * b, lane // 4
            for k in range(R2):
                add("col s2 r", za(x + k * R1, col), allon, False)
                add("col s2 w", za(x + k * R1, col), allon, True)
        if wave < WI:
            row_pass("irow", wave * LI)
            col0 = wave * LI
            col, x = col0 + lane % 8, o1[lane // 8]
            for k in range(R1):
                r = x + k * R2
                rr = np.where((r == 0) | (r == H), 0, np.where(r < H, r, N - r))
                add("icol s1 r", za(rr, col), allon, False)
                add("icol s1 r", za(rr, col + H), allon, False)
                add("icol s1 w", za(x * R1 + k, col), allon, True)
            for b in range(2):
                col, x = col0 + lane % 4 + 4 * b, lane // 4
                for k in range(R2):
                    add("icol s2 r", za(x + k * R1, col), allon, False)
                    add("icol s2 w", za(x + k * R1, col), allon, True)
    for i in range(4):
        for wave in range(WAVES):
            tid = wave * 64 + lane
            u, vr = tid % 64, tid // 64
            for uu in range(2):
                v = 1 + vr + i * 16
                on = v < H
                vv = np.minimum(v, H - 1)
                uc = u + uu * 64
                add("xpow r", za(vv, uc), on, False)
                add("xpow r", za(N - vv, (N - uc) % N), on, False)
                add("xpow w", za(vv, uc), on, True)
    return tot


if __name__ == "__main__":
    for new in (False, True):
        d = evaluate(new)
        # LDS-array cycles: a conflict-free read group costs 1, a write group 1 (MI355X_MICROARCH.md)
        t = sum(v[0] for v in d.values()); i = sum(v[1] for v in d.values())
        print("r03 layout" if new else "pitch 136 ", "cycles/ideal %.3f" % (t / i),
              {k: round(float(v[0] / v[1]), 3) for k, v in d.items() if v[0] > v[1]})

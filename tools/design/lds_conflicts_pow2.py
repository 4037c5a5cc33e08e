"""LDS bank-conflict model of pc_kernel.hip (N = 32 / 64 / 128): same model as lds_conflicts_120.py
(ds_read_b64: 32-lane groups, addresses distinct mod 32; ds_write_b64: 16-lane groups, distinct mod 16; 8-byte units)."""
import sys
import numpy as np
from lds_conflicts_120 import cost

CFG = {32: dict(R1=8, R2=4, SK=3, PITCH=36, T=64), 64: dict(R1=8, R2=8, SK=3, PITCH=72, T=256),
       128: dict(R1=16, R2=8, SK=4, PITCH=136, T=1024)}


def evaluate(N, pitch=None, sk=None, T=None, za_fn=None, row_s2_perm=False, col_s2_blocked=False):
    c = CFG[N]
    R1, R2 = c["R1"], c["R2"]
    P = pitch or c["PITCH"]
    SK = c["SK"] if sk is None else sk
    T = T or c["T"]
    WAVES = T // 64
    LPW = N // WAVES
    H = N // 2
    BMIN = min(R1, R2)
    LI = max(LPW // 2, 64 // BMIN)
    WI = H // LI
    za = za_fn if za_fn is not None else (lambda r, cc: r * P + cc + (cc >> SK))
    lane = np.arange(64)
    allon = np.ones(64, bool)
    tot = {}

    def add(name, addrs, active, write):
        t, i = cost(np.asarray(addrs), active, write)
        a = tot.setdefault(name, [0, 0])
        a[0] += t
        a[1] += i

    def row_pass(tag, line0, LINES):
        for b in range(LINES * R2 // 64):
            q = lane + 64 * b
            line, x = line0 + q // R2, q % R2
            for k in range(R1):
                add(tag + " s1 r", za(line, x + k * R2), allon, False)
                add(tag + " s1 w", za(line, x * R1 + k), allon, True)
        for b in range(LINES * R1 // 64):
            q = lane + 64 * b
            line, x = line0 + q // R1, q % R1
            if row_s2_perm:  # 8 consecutive x of 8 / (R1 / 8) lines per 32-lane group; x depends on the lane only
                lpi = 8 // (R1 // 8)
                line = line0 + ((lane >> 3) % lpi) + lpi * b
                x = (lane & 7) + 8 * (lane // (8 * lpi))
            for k in range(R2):
                add(tag + " s2 r", za(line, x + k * R1), allon, False)
                add(tag + " s2 w", za(line, x + k * R1), allon, True)

    CPR = N // 16
    for wave in range(WAVES):
        row, col = wave * LPW + lane // CPR, (lane % CPR) * 16
        on = lane < LPW * CPR
        for i in range(16):
            add("load w", za(row, col + i), on, True)
        row_pass("row", wave * LPW, LPW)
        col0 = wave * LPW
        CW = 64 // R2
        for b in range(LPW * R2 // 64):
            col, x = col0 + lane % CW + CW * b, lane // CW
            for k in range(R1):
                add("col s1 r", za(x + k * R2, col), allon, False)
                add("col s1 w", za(x * R1 + k, col), allon, True)
        CW = 64 // R1
        for b in range(LPW * R1 // 64):
            col, x = col0 + lane % CW + CW * b, lane // CW
            if col_s2_blocked:  # 8 columns x 8 consecutive x per iteration: x = lane / 8 + 8 (b % (R1/8))
                xb = R1 // 8
                col, x = col0 + (lane & 7) + 8 * (b // xb), (lane >> 3) + 8 * (b % xb)
            for k in range(R2):
                add("col s2 r", za(x + k * R1, col), allon, False)
                add("col s2 w", za(x + k * R1, col), allon, True)
        if wave < WI:
            row_pass("irow", wave * LI, LI)
            col0 = wave * LI
            CW = 64 // R2
            for b in range(LI * R2 // 64):
                col, x = col0 + lane % CW + CW * b, lane // CW
                for k in range(R1):
                    r = x + k * R2
                    rr = np.where((r == 0) | (r == H), 0, np.where(r < H, r, N - r))
                    add("icol s1 r", za(rr, col), allon, False)
                    add("icol s1 r", za(rr, col + H), allon, False)
                    add("icol s1 w", za(x * R1 + k, col), allon, True)
            CW = 64 // R1
            for b in range(LI * R1 // 64):
                col, x = col0 + lane % CW + CW * b, lane // CW
                if col_s2_blocked:
                    xb = R1 // 8
                    col, x = col0 + (lane & 7) + 8 * (b // xb), (lane >> 3) + 8 * (b % xb)
                for k in range(R2):
                    add("icol s2 r", za(x + k * R1, col), allon, False)
                    add("icol s2 w", za(x + k * R1, col), allon, True)
    UPW = min(N, 64)
    RPI = T // UPW
    for i in range((H - 1 + RPI - 1) // RPI):
        for wave in range(WAVES):
            tid = wave * 64 + lane
            u, vr = tid % UPW, tid // UPW
            for uu in range(N // UPW):
                v = 1 + vr + i * RPI
                on = v < H
                vv = np.minimum(v, H - 1)
                uc = u + uu * UPW
                add("xpow r", za(vv, uc), on, False)
                add("xpow r", za(N - vv, (N - uc) % N), on, False)
                add("xpow w", za(vv, uc), on, True)
    return tot


if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    d = evaluate(N)
    t = sum(v[0] for v in d.values()); i = sum(v[1] for v in d.values())
    print("N", N, "current: cycles/ideal %.3f" % (t / i), {k: round(float(v[0] / v[1]), 2) for k, v in d.items() if v[0] > v[1]})
    res = []
    for P in range(N, N + 41):
        for sk in (2, 3, 4, 5, 9):
            if P < N + ((N - 1) >> sk):
                continue
            dd = evaluate(N, P, sk)
            tt = sum(v[0] for v in dd.values()); ii = sum(v[1] for v in dd.values())
            res.append((tt / ii, P, sk))
    res.sort()
    for r in res[:6]:
        print("  %.3f pitch %d skew %d" % r)

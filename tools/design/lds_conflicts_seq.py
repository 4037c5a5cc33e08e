"""LDS bank model of the sequence kernel's HALF tile (pc_seq_kernel.hip): N/2 physical rows; a physical row holds one complex
line of N elements = the two logical half-spectrum rows 2j | 2j+1. Same cost model as lds_conflicts_120.py."""
import sys
import numpy as np
from lds_conflicts_120 import cost

CFG = {64: dict(R1=8, R2=8, SK=3, GAP=12, P=88, W=4), 128: dict(R1=16, R2=8, SK=4, GAP=12, P=152, W=8)}


def evaluate(N, P=None, GAP=None):
    c = CFG[N]
    R1, R2, SK, W = c["R1"], c["R2"], c["SK"], c["W"]
    P = P or c["P"]
    GAP = c["GAP"] if GAP is None else GAP
    H = N // 2
    pcol = lambda cc: cc + (cc >> SK) + np.where(cc >= H, GAP, 0)
    line = lambda j, x: j * P + pcol(x)
    spec = lambda r, cc: (r >> 1) * P + pcol(cc + H * (r & 1))
    out = lambda y1, cc: np.where(cc < H, spec(y1, cc % H), spec(y1 + H, cc % H))
    lane = np.arange(64)
    allon = np.ones(64, bool)
    tot = {}

    def add(name, addrs, active, write):
        t, i = cost(np.asarray(addrs), active, write)
        a = tot.setdefault(name, [0, 0])
        a[0] += t
        a[1] += i

    for wave in range(W):
        l0 = 8 * wave
        # store: lane (line = lane>>3, chunk = lane&7), N/8 elements each
        for i in range(N // 8):
            add("store w", line(l0 + (lane >> 3), (N // 8) * (lane & 7) + i), allon, True)
        # row pass over 8 lines
        for b in range(8 * R2 // 64):
            q = lane + 64 * b
            ln, x = l0 + q // R2, q % R2
            for k in range(R1):
                add("row s1 r", line(ln, x + k * R2), allon, False)
                add("row s1 w", line(ln, x * R1 + k), allon, True)
        for b in range(8 * R1 // 64):
            q = lane + 64 * b
            ln, x = l0 + q // R1, q % R1
            for k in range(R2):
                add("row s2 r", line(ln, x + k * R1), allon, False)
                add("row s2 w", line(ln, x + k * R1), allon, True)
        # untangle: lane (line, ug), u = ug + 8 m
        for m in range(H // 8):
            u = (lane & 7) + 8 * m
            add("unt r", line(l0 + (lane >> 3), u), allon, False)
            add("unt r", line(l0 + (lane >> 3), (N - u) % N), allon, False)
            add("unt w", line(l0 + (lane >> 3), u), allon, True)
            add("unt w", line(l0 + (lane >> 3), u + H), allon, True)
        # column passes over 8 columns: s1 lane (col = lane&7, x = lane>>3); s2 as K1: CW = 64/R1 columns per b
        c0 = 8 * wave
        for rep in range(2):  # forward, inverse
            col, x = c0 + (lane & 7), lane >> 3
            for k in range(R1):
                add("col s1 r", spec(x + k * R2, col), allon, False)
                add("col s1 w", spec(x * R1 + k, col), allon, True)
            CW = 64 // R1
            for b in range(8 * R1 // 64):
                col, x = c0 + lane % CW + CW * b, lane // CW
                for k in range(R2):
                    add("col s2 r", spec(x + k * R1, col), allon, False)
                    add("col s2 w", spec(x + k * R1, col), allon, True)
        # row pairs: s1 lane (pair = lane>>3, x = lane&7)
        y1, x = 8 * wave + (lane >> 3), lane & 7
        for k in range(R1):
            u = x + k * R2
            uu = np.where(u < H, u, np.where(u == H, 0, N - u))
            add("inv s1 r", spec(y1, uu), allon, False)
            add("inv s1 r", spec(y1 + H, uu), allon, False)
            add("inv s1 w", out(y1, x * R1 + k), allon, True)
        for b in range(8 * R1 // 64):
            q = lane + 64 * b
            y1, x = 8 * wave + q // R1, q % R1
            for k in range(R2):
                add("inv s2 r", out(y1, x + k * R1), allon, False)
                add("inv s2 w", out(y1, x + k * R1), allon, True)
    return tot


if __name__ == "__main__":
    for N in (64, 128):
        d = evaluate(N)
        t = sum(v[0] for v in d.values()); i = sum(v[1] for v in d.values())
        print("N", N, "cycles/ideal %.3f" % (t / i), {k: round(float(v[0] / v[1]), 2) for k, v in d.items() if v[0] > v[1]})
        if len(sys.argv) > 1:
            res = []
            for P in range(CFG[N]["P"] - 8, CFG[N]["P"] + 25):
                for G in range(0, 33, 4):
                    H = N // 2
                    if P < N + ((N - 1) >> CFG[N]["SK"]) + G + 1:
                        continue
                    dd = evaluate(N, P, G)
                    tt = sum(v[0] for v in dd.values()); ii = sum(v[1] for v in dd.values())
                    res.append((tt / ii, P, G))
            res.sort()
            print("  best:", res[:8])

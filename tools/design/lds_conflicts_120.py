"""LDS bank-conflict model of the N = 120 K1 kernel (pc_kernel_mixed.hip), used to choose the tile pitch / skew.

Model (MI355X_MICROARCH.md, LDS table): ds_read_b64 is served in two groups of 32 lanes over 64 dword banks, i.e. the
8-byte addresses of a group must be distinct mod 32; ds_write_b64 in four groups of 16 contiguous lanes over 32 banks
(distinct mod 16). Cost of a group = max multiplicity of a bank (identical addresses broadcast on reads).
"""
import sys
import numpy as np

N, H, R1, R2, LPW, WAVES = 120, 60, 15, 8, 8, 15


def cost(addrs, active, write):
    gsz, mod = (16, 16) if write else (32, 32)
    total = ideal = 0
    for s in range(0, 64, gsz):
        a = addrs[s:s + gsz][active[s:s + gsz]]
        if a.size == 0:
            continue
        a = np.unique(a) if not write else a
        total += np.bincount(a % mod, minlength=mod).max()
        ideal += 1
    return total, ideal


def evaluate(za, row_s2_blocked=False):
    lane = np.arange(64)
    tot = {}

    def add(name, addrs, active, write):
        t, i = cost(addrs, active, write)
        a = tot.setdefault(name, [0, 0])
        a[0] += t
        a[1] += i

    allon = np.ones(64, bool)
    for wave in range(WAVES):
        line0 = wave * LPW
        # load
        for b in range(2):
            q = lane + 64 * b
            on = q < LPW * (N // 8)
            row, col = line0 + q // (N // 8), (q % (N // 8)) * 8
            for i in range(8):
                add("load w", za(row, col + i), on, True)
        # row pass (forward: all 15 waves; inverse: lines < 60 -> same pattern, fewer waves)
        line, x = line0 + lane // R2, lane % R2
        for k in range(R1):
            add("row s1 r", za(line, x + k * R2), allon, False)
            add("row s1 w", za(line, x * R1 + k), allon, True)
        for b in range(2):
            q = lane + 64 * b
            ln, xx = line0 + q // R1, q % R1
            on = q < LPW * R1
            if row_s2_blocked:  # lane -> (line = lane / 8, x = 8 b + lane % 8): the stage-1 mapping again
                ln, xx = line0 + lane // 8, 8 * b + lane % 8
                on = xx < R1
                xx = np.minimum(xx, R1 - 1)
            for k in range(R2):
                add("row s2 r", za(ln, xx + k * R1), on, False)
                add("row s2 w", za(ln, xx + k * R1), on, True)
        # forward column pass
        col, x = line0 + lane % LPW, lane // LPW
        for k in range(R1):
            add("col s1 r", za(x + k * R2, col), allon, False)
            add("col s1 w", za(x * R1 + k, col), allon, True)
        for b in range(2):
            q = lane + 64 * b
            cc, xx = line0 + q % LPW, q // LPW
            on = xx < R1
            for k in range(R2):
                add("col s2 r", za(np.minimum(xx, R1 - 1) + k * R1, cc), on, False)
                add("col s2 w", za(np.minimum(xx, R1 - 1) + k * R1, cc), on, True)
        # inverse column pass (column pairs), waves with col0 < 60
        if line0 < H:
            col, x = line0 + lane % LPW, lane // LPW
            on = col < H
            for k in range(R1):
                r = x + k * R2
                rr = np.where((r == 0) | (r == H), 0, np.where(r < H, r, N - r))
                add("inv s1 r", za(rr, col), on, False)
                add("inv s1 r", za(rr, col + H), on, False)
                add("inv s1 w", za(x * R1 + k, col), on, True)
            for b in range(2):
                q = lane + 64 * b
                cc, xx = line0 + q % LPW, q // LPW
                on = (xx < R1) & (cc < H)
                for k in range(R2):
                    add("inv s2 r", za(np.minimum(xx, R1 - 1) + k * R1, cc), on, False)
                    add("inv s2 w", za(np.minimum(xx, R1 - 1) + k * R1, cc), on, True)
    # cross-power: thread g -> (v = 1 + g / 120, u = g % 120), 960 threads
    T = WAVES * 64
    for it in range(-(-(H - 1) * N // T)):
        for wave in range(WAVES):
            g = wave * 64 + lane + T * it
            on = g < (H - 1) * N
            gg = np.minimum(g, (H - 1) * N - 1)
            v, u = 1 + gg // N, gg % N
            add("xpow r", za(v, u), on, False)
            add("xpow r", za(N - v, (N - u) % N), on, False)
            add("xpow w", za(v, u), on, True)
    return tot


def summarize(tot):
    t = sum(v[0] for v in tot.values())
    i = sum(v[1] for v in tot.values())
    return t, i


if __name__ == "__main__":
    best = []
    for pitch in range(120, 171):
        for sk in (None, 3, 4):
            za = (lambda r, c, p=pitch, s=sk: r * p + c + ((c >> s) if s is not None else 0))
            if (N - 1) * pitch + N + (N >> (sk or 9)) > 20000:  # 160 KB of cf
                continue
            t, i = summarize(evaluate(za))
            best.append((t / i, pitch, sk))
    best.sort()
    for r in best[:8]:
        print("cycles/ideal %.3f  pitch %d  col-skew %s" % r)
    cur = evaluate(lambda r, c: r * 121 + c)
    print("current (pitch 121):", "%.3f" % (summarize(cur)[0] / summarize(cur)[1]))
    for k, v in cur.items():
        print("  %-10s %.2f" % (k, v[0] / v[1]))
    if len(sys.argv) > 1:
        p = int(sys.argv[1])
        d = evaluate(lambda r, c: r * p + c)
        for k, v in d.items():
            print("  pitch %d %-10s %.2f" % (p, k, v[0] / v[1]))

#!/usr/bin/env python3
"""Bank model of the FUSED half-tile kernel (csrc/pc_half_kernel.hip as of r05: raw staging, untangle / pairing as stage sources, spectra in
registers) with one more degree of freedom than tools/design/half_banks.py: the ORDER in which a stage's lane groups take the wave's lines.
A stage of pc_plan.hpp (stage_rt / stage_rt_ng) lays lpg lines side by side in a wave; which of the group's lines sub-group `sub` of the
lanes takes is free (every stage is in place per line), and it decides which lines meet inside a 16-lane write group / a 32-lane read half.
For every (pitch, skew) that keeps the workgroups per CU the script picks, per pass and stage, the best permutation of a small family
(identity, bit permutations and XOR masks for power-of-two lpg, multipliers a * sub mod lpg) and prints the LDS cycles per patch pair
against the conflict-free count and against the layout the kernel uses now.
Access model: planned_banks.cycles (MI355X_MICROARCH.md, LDS): ds_read_b64 two halves of 32 lanes over 64 banks; ds_write_b64 four
groups of 16 lanes over 32 banks, at least 6 cycles; ds_read_u16 (raw pixels) 64 lanes over 64 banks.
usage: tools/design/half_lanes.py [M ...] [--pitch P --skew S]"""
import itertools
import sys

from half_banks import chain, plan
from planned_banks import slots

CAP = 160 * 1024


def cyc(addrs, kind):
    """addrs: per lane a tuple of dword indices (or None)."""
    if all(a is None for a in addrs):
        return 0, 0
    if kind == "w64":
        ng, nb, floor = 4, 32, 6
    elif kind == "r64":
        ng, nb, floor = 2, 64, 2
    else:  # r16 / r32: one group of 64 lanes
        ng, nb, floor = 1, 64, 1
    tot = 0
    per = 64 // ng
    for gi in range(ng):
        banks = {}
        for a in addrs[gi * per:(gi + 1) * per]:
            if a is None:
                continue
            for d in a:
                banks.setdefault(d % nb, set()).add(d)
        tot += max((len(v) for v in banks.values()), default=0)
    return max(floor, tot), floor


def b64(unit):
    return (2 * unit, 2 * unit + 1)


class Tile:
    def __init__(self, m, P, skew, shift=3):
        self.m, self.H, self.P, self.P2, self.skew, self.shift = m, m // 2, P, P // 2, skew, shift

    def sk(self, v):
        return (v >> self.shift) if self.skew else 0

    def rows_at(self, l, e):
        return l * self.P + e + self.sk(e)

    def cols_at(self, l, o):  # column l, row o of the spec layout
        return l + self.sk(l) + o * self.P2


def perms(lpg):
    out = {"id": list(range(lpg))}
    if lpg & (lpg - 1) == 0 and lpg > 1:
        nb = lpg.bit_length() - 1
        for bp in itertools.permutations(range(nb)):
            for xm in range(lpg):
                p = []
                for s in range(lpg):
                    t = 0
                    for i in range(nb):
                        t |= ((s >> i) & 1) << bp[i]
                    p.append(t ^ xm)
                out[f"bits{bp}^{xm}"] = p
    for a in range(2, lpg):
        if all(a % q or lpg % q for q in range(2, lpg + 1)):
            out[f"x{a}"] = [(a * s) % lpg for s in range(lpg)]
    if lpg <= 6:
        for i, p in enumerate(itertools.permutations(range(lpg))):
            out[f"p{i}"] = list(p)
    # de-duplicate
    seen, res = set(), {}
    for k, p in out.items():
        if tuple(p) not in seen:
            seen.add(tuple(p))
            res[k] = p
    return res


def stage_cost(t, R, np_, line_fast, line0, nlines, perm, src=None, write_line=None):
    """LDS cycles of one wave's stage over its lines; src: None | 'raw' | 'untangle' | 'pair'."""
    m = t.m
    bpl, SL = m // R, slots(R)
    NB = 16 // SL
    tot = ideal = 0
    at = t.cols_at if line_fast else t.rows_at

    def reads(lanes, j):
        nonlocal tot, ideal
        sets = [[], []]
        kind = "r64"
        for tt in lanes:
            if tt is None:
                sets[0].append(None); sets[1].append(None); continue
            x, l = tt
            e = x + j * bpl
            if src == "raw":
                kind = "r16"
                sets[0].append(((l * t.P * 8 + 2 * e) // 4,)); sets[1].append(None)
            elif src == "untangle":  # column l, row e: Z_{e>>1}[l], Z_{e>>1}[M - l] of the rows layout
                um = t.H if l == 0 else m - l
                sets[0].append(b64(t.rows_at(e >> 1, l))); sets[1].append(b64(t.rows_at(e >> 1, um)))
            elif src == "pair":  # line l, element e: spec rows 2l | 2l + 1 at the mirrored bin
                u = e if e < t.H else (0 if e == t.H else m - e)
                o = u + t.sk(u)
                sets[0].append(b64((2 * l) * t.P2 + o)); sets[1].append(b64((2 * l + 1) * t.P2 + o))
            else:
                sets[0].append(b64(at(l, e))); sets[1].append(None)
        for s in sets:
            c, i = cyc(s, kind)
            tot += c; ideal += i

    def writes(lanes, p):
        nonlocal tot, ideal
        a = []
        for tt in lanes:
            if tt is None:
                a.append(None); continue
            x, l = tt
            if write_line is not None and not write_line(l):
                a.append(None); continue
            k = x % np_
            o = (x - k) * R + k + p * np_
            a.append(b64(at(l, o)))
        c, i = cyc(a, "w64")
        tot += c; ideal += i

    if bpl <= 64:
        lpg = 64 // bpl
        group = NB * lpg
        for g0 in range(0, nlines, group):
            for b in range(NB):
                lanes = []
                for lane in range(64):
                    if line_fast:
                        x, sub = lane // lpg, lane % lpg
                        on = x < bpl
                    else:
                        sub, x = lane // bpl, lane % bpl
                        on = sub < lpg
                    li = g0 + b * lpg + (perm[sub] if on else 0)
                    lanes.append((x, line0 + li) if on and li < nlines else None)
                for j in range(R):
                    reads(lanes, j)
                for p in range(R):
                    writes(lanes, p)
    else:
        for li in range(nlines):
            for b in range(NB):
                lanes = [((lane + 64 * b), line0 + li) if lane + 64 * b < bpl else None for lane in range(64)]
                for j in range(R):
                    reads(lanes, j)
                for p in range(R):
                    writes(lanes, p)
    return tot, ideal


def kernel_cost(m, P, skew, choose=True, fixed=None, shift=3, sample=False):
    """-> (cycles, ideal, {(pass, stage): perm name}) per patch pair, all waves."""
    ch, lpw, waves = plan(m)
    t = Tile(m, P, skew, shift)
    H = m // 2
    wave_list = sorted({0, waves // 2}) if sample else list(range(waves))
    passes = [  # (name, line_fast, first-stage source, multiplicity, last-stage write predicate)
        ("rows", False, "raw", 2, None),
        ("cols_prev", True, "untangle", 1, (lambda l: l == 0)),
        ("cols_cur", True, "untangle", 1, None),
        ("icols", True, None, 1, None),
        ("irows", False, "pair", 1, None),
    ]
    tot = ideal = 0
    picked = {}
    for name, lf, src0, mult, wl in passes:
        np_ = 1
        for si, R in enumerate(ch):
            bpl = m // R
            lpg = 64 // bpl if bpl <= 64 else 1
            fam = perms(lpg) if choose and bpl <= 64 else {"id": list(range(max(lpg, 1)))}
            if fixed and (name, si) in fixed:
                fam = {fixed[(name, si)]: perms(lpg)[fixed[(name, si)]]}
            best = None
            for pn, pm in fam.items():
                c = i = 0
                for w in wave_list:
                    l0 = w * lpw
                    nl = max(0, min(lpw, H - l0))
                    if nl == 0:
                        continue
                    cc, ii = stage_cost(t, R, np_, lf, l0, nl, pm, src=src0 if si == 0 else None,
                                        write_line=wl if si == len(ch) - 1 else None)
                    c += cc; i += ii
                if best is None or c < best[0]:
                    best = (c, i, pn)
            # the same stage code serves cols_prev and cols_cur (one instantiation per sink, but the permutation is the stage's): keep them equal
            tot += best[0] * mult; ideal += best[1] * mult
            picked[(name, si)] = best[2]
            np_ *= R
    return tot, ideal, picked


def fits(m, P, skew, shift=3):
    H = m // 2
    pmin = max(m + (((m - 1) >> shift) if skew else 0), 2 * (H + (((H - 1) >> shift) if skew else 0)))
    return P >= pmin and P % 2 == 0


def wgs(m, P):
    _, _, waves = plan(m)
    return min(CAP // (m // 2 * P * 8 + 8 * m + 192), 32 // waves)


CURRENT = {60: (68, 0), 96: (120, 1), 100: None, 120: (136, 0), 128: (152, 1), 144: (202, 1), 150: (180, 0), 160: (184, 1), 162: (186, 0),
           180: (184, 0), 192: (200, 0), 64: (72, 1)}

if __name__ == "__main__":
    args = [a for a in sys.argv[1:]]
    sizes = [int(a) for a in args if a.isdigit()] or [60, 96, 120, 128, 160]
    for m in sizes:
        ch, lpw, waves = plan(m)
        cur = CURRENT.get(m)
        print(f"M={m} radices={ch} lines/wave={lpw} waves={waves}")
        if cur:
            c0, i0, _ = kernel_cost(m, cur[0], cur[1], choose=False)
            c1, i1, pk = kernel_cost(m, cur[0], cur[1], choose=True)
            print(f"  now   P={cur[0]} skew={cur[1]}: {c0} cycles (ideal {i0}, x{c0 / i0:.2f}); best line orders at this pitch: {c1} (x{c1 / i1:.2f})")
            print("     ", {k: v for k, v in pk.items() if v != 'id'})
            w0 = wgs(m, cur[0])
        else:
            w0 = 1
        res = []
        for skew, shift in ((0, 3), (1, 3), (1, 4), (1, 2), (1, 5)):
            for P in range(m, m + 96, 2):
                if not fits(m, P, skew, shift) or wgs(m, P) < w0:
                    continue
                c, i, pk = kernel_cost(m, P, skew, choose=True, shift=shift, sample=True)
                res.append((c / i, P, skew, shift, pk))
        res.sort(key=lambda r: r[0])
        for r in res[:6]:
            c, i, _ = kernel_cost(m, r[1], r[2], choose=True, fixed=r[4], shift=r[3])
            cid, _, _ = kernel_cost(m, r[1], r[2], choose=False, shift=r[3])
            print(f"  cand  P={r[1]} skew={r[2]} shift={r[3]}: {c} cycles x{c / i:.3f} (identity order: {cid})", {k: v for k, v in r[4].items() if v != 'id'}, flush=True)

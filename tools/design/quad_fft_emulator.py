"""Lane-level emulator of the quad-per-line K1 design for N = 64 (tools/design; not shipped, not a test oracle).

Every 1-D transform is split 64 = 4 (across the four lanes of a quad, exchanged with DPP quad_perm) x 16 (in the
lane's registers), so a 1-D pass costs one LDS read and one LDS write of the tile instead of two of each. The
emulator mirrors the kernel's data flow register by register -- lane maps, sign tricks, the Hermitian packing -- and
is checked against numpy.fft, so that index mistakes are found here and not on the GPU.
"""
import numpy as np

N, H = 64, 32
W = lambda n, k: np.exp(-2j * np.pi * k / n)
K1 = [0, 2, 1, 3]  # lane-in-quad -> radix-4 output index (bit reversed)


def dft16(v):  # natural order in, natural order out, last axis = 16 registers
    return np.fft.fft(v, axis=-1)


def quad_xor(vals, m):  # vals [..., 64 lanes, R]; DPP quad_perm: lane <- lane ^ m
    idx = np.arange(64) ^ m
    return vals[..., idx, :]


def xlane_radix4(vals, s_a, s_b, first, second):
    """own + s*partner twice (v_fmac with a DPP operand), with the i-rotation of lane 3 in between."""
    q = np.arange(64) & 3
    t = vals + np.array(s_a)[q][:, None] * quad_xor(vals, first)
    rot = (q == 3)[:, None]
    t = np.where(rot, 1j * t, t)                     # (x, y) -> (-y, x) on lane 3
    return t + np.array(s_b)[q][:, None] * quad_xor(t, second)


SIGMA = [1, -1, -1, -1]
# natural lanes in (lane q holds a_q), exchange xor 2 then xor 1: lane q ends with SIGMA[q] * Y[K1[q]]
DIT = dict(s_a=[1, 1, -1, -1], s_b=[1, -1, -1, 1])
# bit-reversed lanes in (lane q holds SIGMA[q] * a_{K1[q]}), exchange xor 1 then xor 2: lane q ends with Y[q]
DIF = dict(s_a=[-1, 1, 1, -1], s_b=[-1, -1, 1, 1])
TW = np.array([[SIGMA[qq] * W(64, j * K1[qq]) for j in range(16)] for qq in range(4)])  # ONE table for both kinds


def pass_blocked_in(vals):
    """lane q holds x[16 q + j] in register j  ->  lane q holds X[K1[q] + 4 k2] in register k2."""
    q = np.arange(64) & 3
    t = xlane_radix4(vals, DIT["s_a"], DIT["s_b"], 2, 1)
    return dft16(t * TW[q])


def pass_interleaved_in(vals):
    """lane q holds x[K1[q] + 4 j] in register j  ->  lane q holds X[k2 + 16 q] in register k2."""
    q = np.arange(64) & 3
    return xlane_radix4(dft16(vals) * TW[q], DIF["s_a"], DIF["s_b"], 1, 2)


def cross_power(zk, zm, real_only):
    eps = np.float64(np.finfo(np.float32).eps)
    A = 0.5 * (zk + np.conj(zm))
    B = -0.5j * (zk - np.conj(zm))
    if real_only:
        p = A.real * B.real
        return p / (p * p + eps) + 0j
    P = A * np.conj(B)
    m = abs(P)
    return P * m / (m * m + eps)


def emulate(cur, prev):
    z0 = cur.astype(np.float64) + 1j * prev.astype(np.float64)
    tile = np.zeros((N, N), complex)
    lane = np.arange(64)
    g, q = lane >> 2, lane & 3

    # ---- A: load + forward rows (wave w, quad g -> row 16 w + g; lane q holds columns 16 q + j)
    for w in range(4):
        r = 16 * w + g
        vals = np.stack([z0[r, 16 * q + j] for j in range(16)], axis=-1)          # [64][16]
        out = pass_blocked_in(vals)
        for k2 in range(16):
            tile[r, np.array(K1)[q] + 4 * k2] = out[:, k2]
    assert np.allclose(tile, np.fft.fft(z0, axis=1))

    # ---- B: forward columns (quad -> column 4 w + (g & 3) + 16 (g >> 2); lane q holds rows q + 4 j)
    new = np.zeros_like(tile)
    for w in range(4):
        u = 4 * w + (g & 3) + 16 * (g >> 2)
        vals = np.stack([tile[np.array(K1)[q] + 4 * j, u] for j in range(16)], axis=-1)
        out = pass_interleaved_in(vals)
        for k2 in range(16):
            new[k2 + 16 * q, u] = out[:, k2]
    tile = new
    Z = np.fft.fft2(z0)
    assert np.allclose(tile, Z)

    # ---- C: cross-power + forward column transform of D = conj(C) for columns 0..31 (column 0 packs 0 and 32)
    G = np.zeros((N, H), complex)
    for w in range(2):
        u = (g & 3) + 16 * ((g >> 2) & 1) + 4 * (g >> 3) + 8 * w
        active = u != 0
        vals = np.zeros((64, 16), complex)
        for j in range(16):
            v = j + 16 * q
            c = np.array([cross_power(tile[vv, uu], tile[(N - vv) % N, (N - uu) % N], False) for vv, uu in zip(v, u)])
            vals[:, j] = np.conj(c)
        out = pass_blocked_in(vals)
        for k2 in range(16):
            y = np.array(K1)[q] + 4 * k2
            G[y[active], u[active]] = out[active, k2]
    # special quad (its own wave in the kernel): lanes 0..3 of a wave
    vals = np.zeros((64, 16), complex)
    for j in range(16):
        for qq in range(4):
            v = j + 16 * qq
            ro = v in (0, H)
            c0 = cross_power(tile[v, 0], tile[(N - v) % N, 0], ro)
            ch = cross_power(tile[v, H], tile[(N - v) % N, H], ro)
            vals[qq, j] = np.conj(c0) + 1j * np.conj(ch)
    out = pass_blocked_in(vals)
    for k2 in range(16):
        for qq in range(4):
            G[K1[qq] + 4 * k2, 0] = out[qq, k2]

    # reference for G: column transform of conj(C)
    C = np.zeros((N, N), complex)
    for v in range(N):
        for uu in range(N):
            ro = (v in (0, H)) and (uu in (0, H))
            C[v, uu] = cross_power(Z[v, uu], Z[(N - v) % N, (N - uu) % N], ro)
    Gref = np.fft.fft(np.conj(C), axis=0)
    assert np.allclose(G[:, 1:], Gref[:, 1:H])
    assert np.allclose(G[:, 0], Gref[:, 0] + 1j * Gref[:, H])

    # ---- D: final rows, two rows per complex transform (quad c = 16 w + g -> rows c and c + 32)
    surf = np.zeros((N, N))
    for w in range(2):
        y1, y2 = 16 * w + g, 16 * w + g + 32
        E = np.zeros((64, 16), complex)
        Fm = np.zeros((64, 9), complex)  # Fm[i] = E[N - u_i] for the lane's lower-half columns u_i = K1[q] + 4 i; Fm[8] = E[32]
        for j in range(8):
            u = np.array(K1)[q] + 4 * j
            a, b = G[y1, u], G[y2, u]
            e = a + 1j * b
            f = np.conj(a) + 1j * np.conj(b)
            if j == 0:  # lane 0: packed column
                e = np.where(q == 0, a.real + 1j * b.real, e)
                Fm[:, 8] = a.imag + 1j * b.imag            # only meaningful on lane 0
            E[:, j] = e
            Fm[:, j] = f
        # M[i] = value for the receiver's slot j' = 15 - i: Fm[i] for q != 0, Fm[i + 1] for q == 0
        M = np.stack([np.where(q == 0, Fm[:, i + 1], Fm[:, i]) for i in range(8)], axis=-1)
        src = (g << 2) | np.array([0, 1, 3, 2])[q]       # quad_perm [0,1,3,2]: column classes 1 <-> 3 sit on lanes 2 <-> 3
        R = M[src]
        for i in range(8):
            E[:, 15 - i] = R[:, i]
        out = pass_interleaved_in(E)
        for k2 in range(16):
            x = k2 + 16 * q
            surf[y1, x] = out[:, k2].real
            surf[y2, x] = out[:, k2].imag
    ref = np.fft.ifft2(C).real * N * N
    assert np.allclose(surf, ref, atol=1e-6 * N * N), np.abs(surf - ref).max()
    return surf


def bank_check(pitch=68):
    """ds_read/write_b64: 32 lanes per pass, 64 banks x 4 B -> the 32 addresses (in 8-byte units) must be distinct mod 32."""
    lane = np.arange(64)
    g, q = lane >> 2, lane & 3
    addr = lambda v, u: v * pitch + u + 4 * (v >> 4)
    worst = {}

    def chk(name, a):
        for half in (a[:32], a[32:]):
            c = np.bincount(half % 32, minlength=32).max()
            worst[name] = max(worst.get(name, 1), c)

    for w in range(4):
        for k in range(16):
            chk("A write", addr(16 * w + g, np.array(K1)[q] + 4 * k))
            u = 4 * w + (g & 3) + 16 * (g >> 2)
            chk("B read", addr(np.array(K1)[q] + 4 * k, u))
            chk("B write", addr(k + 16 * q, u))
    for w in range(2):
        u = (g & 3) + 16 * ((g >> 2) & 1) + 4 * (g >> 3) + 8 * w
        for k in range(16):
            v = k + 16 * q
            chk("C read own", addr(v, u))
            chk("C read partner", addr((N - v) % N, (N - u) % N))
            chk("C write", addr(np.array(K1)[q] + 4 * k, u))
        for k in range(8):
            chk("D read y1", addr(16 * w + g, np.array(K1)[q] + 4 * k))
            chk("D read y2", addr(16 * w + g + 32, np.array(K1)[q] + 4 * k))
    return worst


if __name__ == "__main__":
    rng = np.random.default_rng(1)
    prev = rng.integers(0, 256, (N, N)).astype(np.uint8)
    cur = np.roll(prev, (3, -5), axis=(0, 1))
    cur[10:20, 10:20] = rng.integers(0, 256, (10, 10))
    s = emulate(cur, prev)
    k = int(np.argmax(np.fft.fftshift(s)))
    print("peak (shifted):", k % N - H, k // N - H, "  emulator == numpy: OK")
    print("bank conflicts (1 = none):", bank_check())

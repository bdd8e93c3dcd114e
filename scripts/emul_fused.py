"""CPU emulation (numpy, float32 with emulated FMAs) of the fused kernels' FFT arithmetic, pass by pass, beside a
float64 shadow -- to see WHERE the float32 error of a size comes from without spending GPU time.

Not bit-exact with the GPU (hipcc contracts some mul+add pairs the source does not spell out, and v_log_f32 is not
emulated), but the same decomposition, the same twiddle tables and constants, the same operation order: the
statistics of the error are the kernel's.  The structure mirrors scanner_amd/csrc/scn_kernels.hip:
  n = TV*a + M*b + c,  k = p + 16 q + 256 r,  passes 16 x 16 x M   (TV = 16 M)

  python scripts/emul_fused.py [n] [n_buffers] [seed]
"""
import sys

import numpy as np

sys.path.insert(0, ".")
from scanner_amd import synth  # noqa: E402

F = np.float32


def fma(a, b, c):
    return (a.astype(np.float64) * np.float64(b) + np.asarray(c, np.float64)).astype(F)


class C32:
    """complex array as two float32 arrays; every operation rounds like the kernel's scalar f32 code"""

    def __init__(self, x, y):
        self.x, self.y = np.asarray(x, F), np.asarray(y, F)

    def __add__(self, o):
        return C32(self.x + o.x, self.y + o.y)

    def __sub__(self, o):
        return C32(self.x - o.x, self.y - o.y)

    def to64(self):
        return self.x.astype(np.float64) + 1j * self.y.astype(np.float64)


def cmul(a, wx, wy):
    wx, wy = np.asarray(wx, F), np.asarray(wy, F)
    return C32(fma(-a.y, wy, a.x * wx), fma(a.y, wx, a.x * wy))


def mul_w2(a, h):
    return C32((a.x + a.y) * h, (a.y - a.x) * h)


def mul_w6(a, h):
    return C32((a.y - a.x) * h, -(a.x + a.y) * h)


def radix4(x0, x1, x2, x3):
    t0, t1, t2, t3 = x0 + x2, x0 - x2, x1 + x3, x1 - x3
    return t0 + t2, C32(t1.x + t3.y, t1.y - t3.x), t0 - t2, C32(t1.x - t3.y, t1.y + t3.x)


def fft16(v):
    """v: list of 16 C32 (inputs n = 0..15); returns list X[k], k = 0..15"""
    C1, S1, H = F(0.92387953251128675613), F(0.38268343236508977173), F(0.70710678118654752440)
    v = list(v)
    for n0 in range(4):
        v[n0], v[n0 + 4], v[n0 + 8], v[n0 + 12] = radix4(v[n0], v[n0 + 4], v[n0 + 8], v[n0 + 12])
    v[5] = cmul(v[5], C1, -S1)
    v[9] = mul_w2(v[9], H)
    v[13] = cmul(v[13], S1, -C1)
    v[6] = mul_w2(v[6], H)
    v[10] = C32(v[10].y, -v[10].x)
    v[14] = mul_w6(v[14], H)
    v[7] = cmul(v[7], S1, -C1)
    v[11] = mul_w6(v[11], H)
    v[15] = cmul(v[15], -C1, S1)
    for k0 in range(4):
        v[4 * k0], v[4 * k0 + 1], v[4 * k0 + 2], v[4 * k0 + 3] = radix4(v[4 * k0], v[4 * k0 + 1], v[4 * k0 + 2], v[4 * k0 + 3])
    return [v[4 * (k & 3) + (k >> 2)] for k in range(16)]


def stack(lst):  # list of C32 [..] -> C32 with a new leading axis
    return C32(np.stack([c.x for c in lst]), np.stack([c.y for c in lst]))


def unstack(c):
    return [C32(c.x[i], c.y[i]) for i in range(c.x.shape[0])]


def twiddle_table(n):
    a = -2.0 * np.pi * np.arange(n, dtype=np.float64) / n
    return np.cos(a).astype(F), np.sin(a).astype(F)


def fused_fft(xw, M, exact=()):
    """xw: complex64 [B, N] windowed samples, N = 256 M.  Returns the kernel's float32 spectrum as complex128 [B, N] plus the
    per-pass intermediate results.  `exact`: names of steps evaluated in float64 instead ("tw1", "tw2", "p1", "p2", "p3")."""
    B, N = xw.shape
    assert N == 256 * M
    TV = 16 * M
    twx, twy = twiddle_table(N)
    x = C32(xw.real, xw.imag)
    # pass 1: tau in [0, TV): DFT16 over a of x[TV a + tau], times W_N^(tau p)
    xa = C32(x.x.reshape(B, 16, TV), x.y.reshape(B, 16, TV))
    Y = fft16([C32(xa.x[:, a], xa.y[:, a]) for a in range(16)])  # Y[p][B, tau]
    tau = np.arange(TV)
    Y = [Y[0]] + [cmul(Y[p], twx[(tau * p) % N], twy[(tau * p) % N]) for p in range(1, 16)]
    # pass 2: (p, c): DFT16 over b of L1(p, M b + c), times W_(16M)^(c q) = W_N^(16 c q)
    Z = {}
    c = np.arange(M)
    L2 = [[None] * 16 for _ in range(16)]  # L2[p][q] -> [B, c]
    for p in range(16):
        yp = C32(Y[p].x.reshape(B, 16, M), Y[p].y.reshape(B, 16, M))
        z = fft16([C32(yp.x[:, b], yp.y[:, b]) for b in range(16)])  # z[q][B, c]
        for q in range(16):
            L2[p][q] = z[q] if q == 0 else cmul(z[q], twx[(16 * c * q) % N], twy[(16 * c * q) % N])
    # pass 3: for kl = p + 16 q: DFT_M over c -> X[kl + 256 r]
    X = np.zeros((B, N), np.complex128)
    for p in range(16):
        for q in range(16):
            kl = p + 16 * q
            v = L2[p][q]  # [B, M]
            cols = [C32(v.x[:, i], v.y[:, i]) for i in range(M)]
            if M == 16:
                out = fft16(cols)
            elif M == 4:
                out = list(radix4(*cols))
            elif M == 8:
                out = fft8(cols)
            elif M == 32:
                out = dft32_wide(cols)
            elif M == 64:
                out = dft64_wide(cols)
            for r in range(M):
                X[:, kl + 256 * r] = out[r].to64()
    return X


def fft8(z):
    H = F(0.70710678118654752440)
    z = list(z)
    z[0], z[2], z[4], z[6] = radix4(z[0], z[2], z[4], z[6])
    z[1], z[3], z[5], z[7] = radix4(z[1], z[3], z[5], z[7])
    z[3] = mul_w2(z[3], H)
    z[5] = C32(z[5].y, -z[5].x)
    z[7] = mul_w6(z[7], H)
    out = [None] * 8
    for k0 in range(4):
        a, b = z[2 * k0], z[2 * k0 + 1]
        out[k0], out[k0 + 4] = a + b, a - b
    return out


def w_const(r, m):
    return F(np.cos(2 * np.pi * r / m)), F(np.sin(2 * np.pi * r / m))


def dft32_wide(cols):
    ev, od = fft16(cols[0::2]), fft16(cols[1::2])
    out = [None] * 32
    for r in range(16):
        o = od[r]
        if r:
            cr, sr = w_const(r, 32)
            o = cmul(o, cr, -sr)
        out[r], out[r + 16] = ev[r] + o, ev[r] - o
    return out


def dft64_wide(cols):
    """the wide 16384-point kernel's pass 3: lane half e holds c = 4c'+e (va) and c = 4c'+e+2 (vb)"""
    U, V = [None, None], [None, None]
    for e in range(2):
        va, vb = fft16(cols[e::4]), fft16(cols[e + 2::4])
        U[e], V[e] = [None] * 16, [None] * 16
        for r in range(16):
            o = vb[r]
            if r:
                cr, sr = w_const(r, 32)
                o = cmul(o, cr, -sr)
            x0, x1 = va[r] + o, va[r] - o
            if r and e:
                cr, sr = w_const(r, 64)
                x0, x1 = cmul(x0, cr, -sr), cmul(x1, cr, -sr)
            U[e][r], V[e][r] = x0, x1
    out = [None] * 64
    one, mone = F(1.0), F(-1.0)
    for r in range(16):
        u0, u1, v0, v1 = U[0][r], U[1][r], V[0][r], V[1][r]
        out[r] = C32(fma(u1.x, one, u0.x), fma(u1.y, one, u0.y))
        out[r + 32] = C32(fma(u1.x, mone, u0.x), fma(u1.y, mone, u0.y))
        out[r + 16] = C32(fma(v1.y, one, v0.x), fma(v1.x, mone, v0.y))
        out[r + 48] = C32(fma(v1.y, mone, v0.x), fma(v1.x, one, v0.y))
    return out


def fused_fft_16k2(xw, p3_double=True):
    """The product's 16384-point kernel (scn_fft16k2_body): n = 512 a + 32 b + c, k = p + 32 q + 512 r; pass 1 a 32-point DFT
    (two 16-point DFTs + one radix-2 step), pass 2 16 points, pass 3 32 points -- in double (exact here) or float."""
    B, N = xw.shape
    assert N == 16384
    twx, twy = twiddle_table(N)
    x = C32(xw.real, xw.imag)
    xa = C32(x.x.reshape(B, 32, 512), x.y.reshape(B, 32, 512))
    ev = fft16([C32(xa.x[:, 2 * a], xa.y[:, 2 * a]) for a in range(16)])
    od = fft16([C32(xa.x[:, 2 * a + 1], xa.y[:, 2 * a + 1]) for a in range(16)])
    tau = np.arange(512)
    A = [None] * 32
    for p in range(16):
        o = od[p]
        if p:
            cr, sr = w_const(p, 32)
            o = cmul(o, cr, -sr)
        A[p], A[p + 16] = ev[p] + o, ev[p] - o
    A = [A[0]] + [cmul(A[p], twx[(tau * p) % N], twy[(tau * p) % N]) for p in range(1, 32)]
    c = np.arange(32)
    X = np.zeros((B, N), np.complex128)
    L2 = np.zeros((B, 32, 16, 32), np.complex128)  # [p][q][c], float-valued
    for p in range(32):
        yp = C32(A[p].x.reshape(B, 16, 32), A[p].y.reshape(B, 16, 32))
        z = fft16([C32(yp.x[:, b], yp.y[:, b]) for b in range(16)])
        for q in range(16):
            zz = z[q] if q == 0 else cmul(z[q], twx[(32 * c * q) % N], twy[(32 * c * q) % N])
            L2[:, p, q] = zz.to64()
    if p3_double:
        out = np.fft.fft(L2, axis=-1)  # exact 32-point DFT over c of the float exchange values
        for p in range(32):
            for q in range(16):
                X[:, p + 32 * q::512] = out[:, p, q]
    else:
        for p in range(32):
            for q in range(16):
                v = L2[:, p, q]
                o = dft32_wide([C32(v.real[:, i], v.imag[:, i]) for i in range(32)])
                for r in range(32):
                    X[:, p + 32 * q + 512 * r] = o[r].to64()
    return X


def bh_window(n):
    i = np.arange(n, dtype=np.float64) / (n - 1.0)
    return (0.35875 - 0.48829 * np.cos(2 * np.pi * i) + 0.14128 * np.cos(4 * np.pi * i) - 0.01168 * np.cos(6 * np.pi * i)).astype(F)


def metric(P, Pref):
    return np.abs(P - Pref) / np.maximum(Pref, Pref.mean(axis=-1, keepdims=True))


def db32(P):
    """the reference's map, float result: 10*log10(sqrt(P)) rounded to float"""
    return (5.0 * np.log10(P)).astype(F)


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    nb = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 900
    import scipy.fft

    x = synth.cfloat_batch(n, nb, seed=seed)
    w = bh_window(n)
    xw = C32(x.real * w, x.imag * w)  # float products, as VOLK's multiply
    xw64 = xw.to64()
    X64 = np.fft.fft(xw64, axis=-1)
    P64 = np.abs(X64) ** 2
    Xk = fused_fft(xw64.astype(np.complex64), n // 256)
    Xp = scipy.fft.fft(xw64.astype(np.complex64), axis=-1).astype(np.complex128)
    cases = [("16 x 16 x M, float", Xk), ("pocketfft float32", Xp)]
    if n == 16384:
        cases += [("32 x 16 x 32, float pass 3", fused_fft_16k2(xw64.astype(np.complex64), False)),
                  ("32 x 16 x 32, double pass 3", fused_fft_16k2(xw64.astype(np.complex64), True))]
    for name, X in cases:
        P = np.abs(X) ** 2
        m_lin = metric(P, P64).max(axis=-1)
        # through the float dB map on both sides (what the tests compare)
        m_db = metric(10.0 ** (db32(P).astype(np.float64) / 5.0), 10.0 ** (db32(P64).astype(np.float64) / 5.0)).max(axis=-1)
        err = np.abs(X - X64)
        print(f"{name:28s} |dX| rms/|X|max {np.sqrt((err ** 2).mean()) / np.abs(X64).max():.3g}   metric on exact power: median {np.median(m_lin):.3g} p99 "
              f"{np.percentile(m_lin, 99):.3g} max {m_lin.max():.3g}   through float dB: median {np.median(m_db):.3g} max {m_db.max():.3g}")

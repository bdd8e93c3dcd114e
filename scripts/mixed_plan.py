"""Plans of the mixed-radix fused kernels (scn_mixed.hip): for every supported size N = 2^a 3^b 5^c that is not a power of two, the
three radices of its passes and the LDS row pitches, chosen here and written to scanner_amd/csrc/scn_mixed_plans.h.

   python scripts/mixed_plan.py            # verify (numpy emulation of the three passes against numpy.fft) and print the table
   python scripts/mixed_plan.py --write    # ... and regenerate the header
   python scripts/mixed_plan.py --search   # choose radices and pitches again (brute force: minutes)

N = R1 R2 R3, input index n = T1 a + R3 b + c (T1 = R2 R3), output index k = p + R1 q + R1 R2 r:
  pass 1  thread tau = R3 b + c (T1 of them): DFT_R1 over a of x[T1 a + tau] w[..]  -> * W_N^(tau p)      -> L1(p, tau) = P1 p + tau
  pass 2  thread (p, c) (R1 R3 of them):      DFT_R2 over b of L1(p, R3 b + c)      -> * W_(R2 R3)^(c q) -> L2(c, kl) = P2 c + kl, kl = p + R1 q
  pass 3  thread kl (R1 R2 of them):          DFT_R3 over c of L2(c, kl)            -> X[kl + R1 R2 r]
R1 is the smallest radix, so pass 1 has the most threads (one per workgroup thread, its twiddles and window taps in registers
for the whole launch) and passes 2 / 3 run on a leading subset.  The pitches P1 = T1 + pad1, P2 = R1 R2 + pad2 are the ones with
the fewest LDS bank conflicts under the guide's model: a ds_*_b64 is served in groups of 16 consecutive lanes, conflict-free iff
their 8-byte slots are distinct mod 16 (32 banks x 4 bytes)."""
import itertools
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the sizes VERDICT r4 (next 7) names first, and round decimal / 3-smooth sizes between them
# -- and the families a radix-2 user reaches for next: 3 * 2^k, 5 * 2^k, 15 * 2^k --
# (up to 10000: beyond ~10600 every factorisation has a pass of more than 512 threads, see GeoMixed in scn_mixed.hip)
SIZES = [1000, 1200, 1280, 1500, 1536, 1920, 2000, 2400, 2500, 2560, 3000, 3072, 3600, 3840, 4000, 4800, 5000, 5120, 6000, 6144, 7200, 7680,
         8000, 9000, 9600, 10000]
RMAX = 32
# what `--search` found (the pad search is a brute force over 256 pitch pairs per factorisation: minutes in Python), kept here so
# that --write and the verification are instant; N: (R1, R2, R3, pad1, pad2, extra LDS cycles per buffer)
CHOSEN = {
    1000: (10, 10, 10, 6, 2, 50), 1200: (10, 10, 12, 4, 2, 50), 1500: (10, 10, 15, 9, 11, 0), 2000: (10, 10, 20, 12, 1, 70),
    2400: (10, 15, 16, 0, 1, 0), 2500: (10, 10, 25, 15, 5, 0), 3000: (10, 20, 15, 3, 7, 0), 3600: (15, 15, 16, 0, 0, 0),
    4000: (10, 20, 20, 4, 1, 140), 4800: (15, 20, 16, 0, 1, 0), 5000: (10, 20, 25, 5, 1, 0), 6000: (15, 20, 20, 4, 1, 220),
    7200: (18, 20, 20, 4, 1, 260), 8000: (20, 20, 20, 4, 1, 300), 9000: (18, 20, 25, 5, 1, 0), 9600: (20, 20, 24, 8, 1, 200),
    10000: (20, 20, 25, 5, 9, 0), 12000: (20, 24, 25, 1, 9, 0), 12288: (16, 24, 32, 0, 1, 0), 14400: (24, 24, 25, 1, 9, 0),
    15000: (24, 25, 25, 8, 1, 0), 16000: (20, 25, 32, 0, 1, 0),
    1280: (8, 10, 16, 0, 1, 0), 1536: (8, 12, 16, 0, 1, 0), 1920: (10, 12, 16, 0, 1, 0), 2560: (10, 16, 16, 0, 1, 0), 3072: (12, 16, 16, 0, 1, 0),
    3840: (15, 16, 16, 0, 1, 0), 5120: (16, 20, 16, 0, 1, 0), 6144: (16, 24, 16, 0, 1, 0), 7680: (16, 20, 24, 8, 1, 160),
}


# The sizes beyond 10000: two virtual threads per thread and ONE in-place exchange (GeoMixedBig in scn_mixed.hip): pass 2 writes its
# outputs back to the slots it read, pass 3 reads L1(p, R3 q + c).  The SMALLEST radix goes last: pass 3 runs in double there (the
# accuracy tail of these sizes, profiles/r05_experiments.md section 4) and a 16 .. 24-point double DFT fits the registers where a
# 25- or 32-point one does not.  N: (R1, R2, R3, pad1, extra LDS cycles per buffer)
BIG_CHOSEN = {12000: (24, 25, 20, 1, 1140), 12288: (32, 24, 16, 1, 0), 14400: (24, 25, 24, 1, 600), 15000: (25, 25, 24, 1, 1152), 16000: (32, 25, 20, 1, 1200),
              10240: (32, 20, 16, 1, 0), 12800: (32, 20, 20, 1, 960), 15360: (32, 24, 20, 1, 1152)}


def big_cost(r1, r2, r3, pad1):
    t1, p1, v2, v3 = r2 * r3, r2 * r3 + pad1, r1 * r3, r1 * r2
    w = -(-(-(-max(t1, v2, v3) // 2)) // 64) * 64
    cost = 0
    for h in range(2):
        tvs = [t + h * w for t in range(w)]
        for b in range(r2):  # pass-2 reads and its in-place writes
            cost += 2 * conflicts([None if tv >= v2 else (tv // r3) * p1 + r3 * b + tv % r3 for tv in tvs])
        for c in range(r3):  # pass-3 reads
            cost += conflicts([None if tv >= v3 else (tv % r1) * p1 + r3 * (tv // r1) + c for tv in tvs])
    return cost, w


def emulate_big(n, r1, r2, r3, seed=0):
    """the in-place form: the same three DFTs, exchange 2 written back to L1's slots; returns max |X - fft(x)| / max |fft(x)|"""
    rng = np.random.default_rng(seed)
    x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    t1 = r2 * r3
    W = lambda m, e: np.exp(-2j * np.pi * (np.asarray(e) % m) / m)  # noqa: E731
    l1 = np.zeros((r1, t1), complex)
    a, p = np.arange(r1), np.arange(r1)
    for tau in range(t1):
        l1[:, tau] = (W(r1, np.outer(p, a)) @ x[t1 * a + tau]) * W(n, tau * p)
    b, q = np.arange(r2), np.arange(r2)
    for pp in range(r1):
        for c in range(r3):
            l1[pp, r3 * q + c] = (W(r2, np.outer(q, b)) @ l1[pp, r3 * b + c]) * W(r2 * r3, c * q)   # in place: slots R3 q + c
    X = np.zeros(n, complex)
    cc, r = np.arange(r3), np.arange(r3)
    for kl in range(r1 * r2):
        pp, qq = kl % r1, kl // r1
        X[kl + r1 * r2 * r] = W(r3, np.outer(r, cc)) @ l1[pp, r3 * qq + cc]
    ref = np.fft.fft(x)
    return np.abs(X - ref).max() / np.abs(ref).max()


def factorizations(n):
    out = []
    for r1 in range(2, RMAX + 1):
        if n % r1:
            continue
        for r2 in range(r1, RMAX + 1):
            if (n // r1) % r2:
                continue
            r3 = n // r1 // r2
            if r1 <= r3 <= RMAX:
                out.append((r1, r2, r3))
    return out


def smooth235(r):
    for f in (2, 3, 5):
        while r % f == 0:
            r //= f
    return r == 1


def conflicts(slots):
    """extra LDS cycles of one wave instruction: per aligned 16-lane group, (largest number of lanes on one slot residue mod 16) - 1"""
    extra = 0
    for g in range(0, len(slots), 16):
        grp = [s % 16 for s in slots[g:g + 16] if s is not None]
        if grp:
            extra += max(np.bincount(grp, minlength=16)) - 1
    return extra


def conflicts_np(slots, active):
    """conflicts() for many wave instructions at once: slots [instructions, lanes] (lanes a multiple of 16), active [lanes]"""
    res = np.where(active[None, :], slots % 16, -1).reshape(slots.shape[0], -1, 16)
    worst = np.zeros(res.shape[:2], int)
    for r in range(16):
        worst = np.maximum(worst, (res == r).sum(axis=2))
    return int(np.maximum(worst - 1, 0).sum())


def pitch_cost(r1, r2, r3, pad1, pad2):
    t1, p1, p2 = r2 * r3, r2 * r3 + pad1, r1 * r2 + pad2
    threads = -(-t1 // 64) * 64
    t = np.arange(threads)
    active, p, c = t < r1 * r3, t // r3, t % r3
    k = np.arange(r2)[:, None]
    return conflicts_np(p[None, :] * p1 + r3 * k + c[None, :], active) + conflicts_np(c[None, :] * p2 + p[None, :] + r1 * k, active)  # pass-2 reads, pass-2 writes


def choose(n):
    best = None
    for r1, r2, r3 in factorizations(n):
        if not all(smooth235(r) for r in (r1, r2, r3)):
            continue
        for a, b in ((r2, r3), (r3, r2)):  # which of the two larger radices goes last
            t1 = a * b
            if t1 > 512:
                continue
            balance = max(r1, a, b) / r1
            pads = min(((pitch_cost(r1, a, b, p1, p2), p1 + p2, p1, p2) for p1, p2 in itertools.product(range(16), range(16))))
            # idle lanes of passes 2 and 3 (they run on r1*b and r1*a of the t1 threads), then conflicts
            idle = (1 - r1 * b / t1) + (1 - r1 * a / t1)
            key = (round(balance, 3), round(idle, 3), pads[0], max(a, b))
            if best is None or key < best[0]:
                best = (key, (r1, a, b, pads[2], pads[3], pads[0]))
    return best[1] if best else None


def emulate(n, r1, r2, r3, seed=0):
    """the three passes in complex128 with exactly the index maps of the kernel; returns max |X - fft(x)| / max |fft(x)|"""
    rng = np.random.default_rng(seed)
    x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    t1 = r2 * r3
    W = lambda m, e: np.exp(-2j * np.pi * (e % m) / m)  # noqa: E731
    y1 = np.zeros((r1, t1), complex)
    for tau in range(t1):
        v = x[t1 * np.arange(r1) + tau]
        for p in range(r1):
            y1[p, tau] = sum(v[a] * W(r1, a * p) for a in range(r1)) * W(n, tau * p)
    y2 = np.zeros((r3, r1 * r2), complex)
    for p in range(r1):
        for c in range(r3):
            v = y1[p, r3 * np.arange(r2) + c]
            for q in range(r2):
                y2[c, p + r1 * q] = sum(v[b] * W(r2, b * q) for b in range(r2)) * W(r2 * r3, c * q)
    X = np.zeros(n, complex)
    for kl in range(r1 * r2):
        v = y2[:, kl]
        for r in range(r3):
            X[kl + r1 * r2 * r] = sum(v[c] * W(r3, c * r) for c in range(r3))
    ref = np.fft.fft(x)
    return np.abs(X - ref).max() / np.abs(ref).max()


def main():
    rows = []
    for n in SIZES:
        assert n & (n - 1), n
        plan = choose(n) if "--search" in sys.argv else CHOSEN[n]
        assert plan, n
        r1, r2, r3, pad1, pad2, cost = plan
        assert r1 * r2 * r3 == n and r1 <= min(r2, r3) and max(r2, r3) <= RMAX and r2 * r3 <= 512 and cost == pitch_cost(r1, r2, r3, pad1, pad2)
        err = emulate(n, r1, r2, r3) if n <= 3000 or "--all" in sys.argv else None
        assert err is None or err < 1e-12, (n, err)
        t1 = r2 * r3
        threads = -(-t1 // 64) * 64
        exch = max(r1 * (t1 + pad1), r3 * (r1 * r2 + pad2))
        rows.append((n, r1, r2, r3, pad1, pad2))
        print(f"{n:6d} = {r1:2d} x {r2:2d} x {r3:2d}   threads {threads:4d} (pass 1: {t1}, 2: {r1 * r3}, 3: {r1 * r2})  P1 {t1 + pad1:4d} P2 {r1 * r2 + pad2:4d}  "
              f"LDS {exch * 8 / 1024:6.1f} KiB  extra LDS cycles per buffer {cost:3d}" + ("" if err is None else f"  emulated vs numpy.fft {err:.1e}"))
    big_rows = []
    for n, (r1, r2, r3, pad1, cost) in sorted(BIG_CHOSEN.items()):
        c, w = big_cost(r1, r2, r3, pad1)
        assert r1 * r2 * r3 == n and r3 <= min(r1, r2) and c == cost, (n, c)
        err = emulate_big(n, r1, r2, r3)
        assert err < 1e-12, (n, err)
        big_rows.append((n, r1, r2, r3, pad1))
        print(f"{n:6d} = {r1:2d} x {r2:2d} x {r3:2d}   threads {w:4d} x 2 virtual (pass 1: {r2 * r3}, 2: {r1 * r3}, 3: {r1 * r2})  P1 {r2 * r3 + pad1:4d} (in place)  "
              f"LDS {r1 * (r2 * r3 + pad1) * 8 / 1024:6.1f} KiB  extra LDS cycles per buffer {cost:3d}  emulated vs numpy.fft {err:.1e}")
    if "--write" in sys.argv:
        with open(os.path.join(ROOT, "scanner_amd", "csrc", "scn_mixed_plans.h"), "w") as f:
            f.write("// scn_mixed_plans.h -- GENERATED by scripts/mixed_plan.py --write: the sizes of the mixed-radix fused kernels (scn_mixed.hip),\n"
                    "// their three radices (R1 the smallest), the pads of the two LDS row pitches and the translation unit of scn_mixed.hip\n"
                    "// that instantiates the size (build.py compiles eight side by side).  X(N, R1, R2, R3, PAD1, PAD2, UNIT)\n"
                    "#define SCN_MIXED_PLANS(X) \\\n")
            f.write(" \\\n".join(f"  X({n}, {r1}, {r2}, {r3}, {p1}, {p2}, {k % 6})" for k, (n, r1, r2, r3, p1, p2) in enumerate(rows)) + "\n")
            f.write("// the sizes beyond 10000 (GeoMixedBig: two virtual threads per thread, one in-place exchange).  X(N, R1, R2, R3, PAD1, UNIT)\n"
                    "#define SCN_MIXED_BIG_PLANS(X) \\\n")
            f.write(" \\\n".join(f"  X({n}, {r1}, {r2}, {r3}, {p1}, {6 + k % 2})" for k, (n, r1, r2, r3, p1) in enumerate(sorted(big_rows, key=lambda r: -r[0]))) + "\n")
        print("wrote scanner_amd/csrc/scn_mixed_plans.h")


if __name__ == "__main__":
    main()

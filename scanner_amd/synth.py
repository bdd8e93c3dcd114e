"""Deterministic synthetic IQ (the reference has no file or synthetic SignalSource; every
source needs USB hardware -- SURVEY.md section 4).

Recipe: complex Gaussian noise of `sigma` per component plus, per buffer,
0..max_tones complex tones with amplitudes U[0.05, 0.5] at fractional bins U[0, N).
`numpy` flavour for tests / host staging, `torch` flavour to fill large device batches
for the bench without a PCIe copy.  Both are pure functions of (seed, shape).
"""
import numpy as np

from . import capi


def cfloat_batch(n, n_buffers, seed, sigma=0.05, max_tones=4):
    """complex64 [n_buffers, n]"""
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((n_buffers, n, 2), dtype=np.float32)
    x *= np.float32(sigma)
    x = x.view(np.complex64).reshape(n_buffers, n)
    k = np.arange(n, dtype=np.float64)
    n_tones = rng.integers(0, max_tones + 1, size=n_buffers)
    for b in range(n_buffers):
        for _ in range(n_tones[b]):
            amp = rng.uniform(0.05, 0.5)
            f = rng.uniform(0, n)
            ph = rng.uniform(0, 2 * np.pi)
            x[b] += (amp * np.exp(1j * (2 * np.pi * f * k / n + ph))).astype(np.complex64)
    return x


def quantize(x, kind, full_scale=None):
    """complex64 [B, n] -> raw wire format of `kind` (what an SDR front-end would deliver).
    int16: round(clip(2047 * x)) (12-bit ADC, bladeRF SC16_Q11); int8: round(clip(127 * x))."""
    x = np.asarray(x)
    if kind == capi.KIND_FLOAT_COMPLEX:
        return np.ascontiguousarray(x, np.complex64)
    iq = np.stack([x.real, x.imag], axis=-1).astype(np.float64)  # [B, n, 2]
    if kind in (capi.KIND_SHORT_COMPLEX, capi.KIND_SHORT):
        fs = 2047 if full_scale is None else full_scale
        q = np.clip(np.rint(iq * fs), -fs - 1, fs).astype(np.int16)
        if kind == capi.KIND_SHORT:  # planar: I[n] then Q[n]
            q = np.ascontiguousarray(np.moveaxis(q, -1, -2))
        return q
    if kind == capi.KIND_BYTE_COMPLEX:
        fs = 127 if full_scale is None else full_scale
        return np.clip(np.rint(iq * fs), -fs - 1, fs).astype(np.int8)
    raise ValueError(kind)


def cfloat_batch_torch(n, n_buffers, seed, device, sigma=0.05, max_tones=4, chunk=1024):
    """Same recipe generated directly in device memory: float32 tensor [n_buffers, n, 2]."""
    import torch

    g = torch.Generator(device=device)
    g.manual_seed(seed)
    out = torch.empty((n_buffers, n, 2), dtype=torch.float32, device=device)
    k = torch.arange(n, dtype=torch.float32, device=device)
    for lo in range(0, n_buffers, chunk):
        hi = min(n_buffers, lo + chunk)
        b = hi - lo
        x = torch.randn((b, n, 2), generator=g, device=device, dtype=torch.float32) * sigma
        nt = torch.randint(0, max_tones + 1, (b,), generator=g, device=device)
        for tone in range(max_tones):
            amp = torch.rand((b,), generator=g, device=device) * 0.45 + 0.05
            amp = amp * (nt > tone)
            f = torch.rand((b,), generator=g, device=device) * n
            ph = torch.rand((b,), generator=g, device=device) * (2 * np.pi)
            arg = (2 * np.pi / n) * f[:, None] * k[None, :] + ph[:, None]
            x[..., 0] += amp[:, None] * torch.cos(arg)
            x[..., 1] += amp[:, None] * torch.sin(arg)
        out[lo:hi] = x
    return out


# ---- BASELINE config 4: a full-table sweep whose hit list has a closed form -----------------------------------
# FrequencyTable(8e6, 0, 16384*6e6) -> 16384 centres 6 MHz apart (frequencyTable.cpp:9-37), one 4096-pt buffer each.
# Every 4th centre carries one emitter placed EXACTLY on a bin centre of that buffer, so its spectrum is the
# window's DFT sampled at integers: A*|W[m]| in bin i0+m.  With A = 0.5 and Blackman-Harris the bins i0-3..i0+3
# clear a 10 dB threshold (the outermost by 0.78 dB) and i0+-4 are 50 dB below it; the noise (sigma 1e-3 per
# component) peaks ~20 dB under the threshold over a whole sweep and moves the hit powers by < 0.1 dB.
C4_SIGMA = 1e-3
C4_AMPLITUDE = 0.5


def blackman_harris(n):
    """gr::fft::window::build(WIN_BLACKMAN_HARRIS, n, 0.0) as process.cpp:18 calls it: 4-term, symmetric, float."""
    x = np.arange(n, dtype=np.float64) / (n - 1.0)
    return (0.35875 - 0.48829 * np.cos(2 * np.pi * x) + 0.14128 * np.cos(4 * np.pi * x)
            - 0.01168 * np.cos(6 * np.pi * x)).astype(np.float32)


def c4_emitters(n_centres, n, seed=4, use_bandwidth=0.75):
    """(centre index, fftshift-ordered bin i0) of the planted emitters: inside the evaluated band
    (process.cpp:48-52), clear of its edges and of the DC window."""
    rng = np.random.default_rng(seed)
    centres = np.arange(1, n_centres, 4, dtype=np.int64)
    u = int(use_bandwidth * n / 2.0)
    lo, hi = n // 2 - u + 8, n // 2 + u - 8
    i0 = rng.integers(lo, hi + 1, size=centres.size)
    i0[np.abs(i0 - n // 2) < 12] += 24
    return centres, i0.astype(np.int64)


def c4_expected_hits(window, fc_all, centres, i0, n, sample_rate, threshold, guard=0.2):
    """The (seq_id = centre index, i, power_db, freq_hz) list the sweep must report, noise-free closed form, with the
    reference's frequency arithmetic (process.cpp:38-39,55-57; the cast of a negative double follows x86-64)."""
    W = np.abs(np.fft.fft(np.asarray(window, np.float64)))
    db = {m: 10.0 * np.log10(C4_AMPLITUDE * W[m % n]) for m in range(-6, 7)}
    if any(abs(db[m] - threshold) <= guard for m in db):
        raise ValueError("a leakage bin of the planted tones sits on the threshold")
    ms = [m for m in range(-6, 7) if db[m] > threshold]
    out = np.zeros(len(centres) * len(ms), capi.HIT_DTYPE)
    bin_step = int(sample_rate) // n
    k = 0
    for c, b in zip(centres, i0):
        start = fc_all[c] - float(int(sample_rate) // 2)
        for m in ms:
            i = int(b + m)
            f = start + float((i * bin_step) & 0xFFFFFFFF)
            out[k] = (c, i, db[m], np.uint64(np.int64(f)) if f < 0 else np.uint64(f))
            k += 1
    return out


def _c4_tones(n, i0):
    k = ((i0 - n // 2)[:, None] * np.arange(n, dtype=np.int64)[None, :]) % n  # exact phase index
    ang = (2.0 * np.pi / n) * k.astype(np.float64)
    return (C4_AMPLITUDE * np.stack([np.cos(ang), np.sin(ang)], axis=-1)).astype(np.float32)  # [E, n, 2]


def c4_shard(n, first, count, centres, i0, seed):
    """numpy: complex64 [count, n] of table entries [first, first+count) (noise differs from the torch flavour)."""
    rng = np.random.default_rng(seed * 7919 + first)
    x = rng.standard_normal((count, n, 2), dtype=np.float32) * np.float32(C4_SIGMA)
    mine = (centres >= first) & (centres < first + count)
    x[centres[mine] - first] += _c4_tones(n, i0[mine])
    return x.view(np.complex64).reshape(count, n)


def c4_shard_torch(n, first, count, centres, i0, seed, device):
    """The same sweep generated in device memory: float32 [count, n, 2]; the noise is a pure function of
    (seed, first, count), the tones of the emitter list alone."""
    import torch

    g = torch.Generator(device=device)
    g.manual_seed(seed * 7919 + first)
    x = torch.randn((count, n, 2), generator=g, device=device, dtype=torch.float32) * C4_SIGMA
    mine = (centres >= first) & (centres < first + count)
    if mine.any():
        x[torch.from_numpy(centres[mine] - first).to(device)] += torch.from_numpy(_c4_tones(n, i0[mine])).to(device)
    return x

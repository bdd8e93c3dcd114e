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

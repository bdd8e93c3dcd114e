"""Parity at the reference's other sample rates.  Every front-end sets its own rate -- HackRF 8 / 10 / 12.5 / 16 / 20 Msps
(hackRFSource.cpp:156-161), 61.44 Msps on the B210, 2.4 Msps RTL-SDR -- and process_fft derives the reported frequency from it
in integer arithmetic: start = fc - fs / 2 with fs / 2 a uint32 division (process.cpp:38), bin_step = fs / N TRUNCATED to
uint32 (process.cpp:39; 12.5e6 / 8192 -> 1525, an odd rate loses its half Hertz), freq = uint64(start + i * bin_step)
(process.cpp:55-57).  freq_hz, i, the order and the trigger flag are demanded bit for bit from every kernel family of the HIP
path (fused small / 4096 / 8192 / 16384, mixed-radix, four-step, Bluestein), in both output modes, and from the
hand-derived known answers of tests/golden/make_golden.py section 2b."""
import os

import numpy as np
import pytest

from scanner_amd import Plan, capi, synth
from tests import tolerances as tol

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
RATES = [2400000, 10000000, 12500000, 20000000, 61440000, 7999999]
SIZES = [64, 1024, 4096, 8192, 16384, 3000, 32768, 1023]   # one per kernel family (tests/test_dispatch_gpu.py walks every size)


@pytest.fixture(scope="module")
def torch_cuda(built_lib):
    import torch

    assert torch.cuda.is_available(), "-m gpu tests need a GPU; refusing to skip silently"
    return torch


def _dev(torch, raw):
    return torch.from_numpy(np.ascontiguousarray(raw).view(np.uint8).reshape(-1)).cuda()


@pytest.mark.parametrize("fs", RATES)
@pytest.mark.parametrize("n", SIZES)
def test_spectrum_and_hits_at_sample_rate(torch_cuda, oracle_mod, n, fs):
    nb = {64: 40, 1024: 24, 4096: 12, 8192: 8, 16384: 6, 3000: 10, 32768: 4, 1023: 6}[n]
    kind = capi.KIND_SHORT_COMPLEX if n in (8192, 3000) else capi.KIND_FLOAT_COMPLEX
    x = synth.cfloat_batch(n, nb, seed=n + fs % 1000)
    raw = synth.quantize(x, kind)
    # centres as a sweep at this rate would tune them (0.75 fs apart), starting low enough that buffer 0's start frequency
    # fc - fs / 2 is NEGATIVE (the cast of a negative double to uint64, process.cpp:57, follows x86-64) ...
    fc = 0.25 * fs + 0.75 * fs * np.arange(nb)
    fc[-1] = 5.9e9 + 0.5                        # ... and one beyond 2^32 Hz with a fractional part
    seq = np.arange(77, 77 + nb, dtype=np.uint64)
    o_all = oracle_mod.Oracle(n, fs, 1e9, kind=kind)
    p_ref, _, _ = o_all.run(raw, threads=4)
    thr = tol.pick_threshold(p_ref, n, start=8.0 if n >= 1024 else 2.0)
    p_ref, h_ref, t_ref = oracle_mod.Oracle(n, fs, thr, kind=kind).run(raw, fc, seq, threads=4)
    assert len(h_ref) > 0, "the case must report detections"
    step, half = fs // n, fs // 2
    for flags in (capi.OUT_SPECTRUM | capi.OUT_HITS, capi.OUT_HITS):
        with Plan(n, fs, thr, kind=kind, max_batch=nb, max_hits=nb * n, flags=flags) as plan:
            plan.submit_device(0, _dev(torch_cuda, raw), nb, fc, seq)
            p, h, t = plan.collect(0, hit_cap=nb * n, want_power=bool(flags & capi.OUT_SPECTRUM))
        if flags & capi.OUT_SPECTRUM:
            tol.compare_spectra(p, p_ref)
        assert len(h) == len(h_ref)
        for f in ("seq_id", "i", "freq_hz"):
            assert np.array_equal(h[f], h_ref[f]), (f, flags)
        assert np.array_equal(t, t_ref)
        # and the arithmetic itself, restated: uint64(double(fc) - fs/2 + double(uint32(i * bin_step)))
        b = (h["seq_id"] - 77).astype(np.int64)
        want = (fc[b] - float(half)) + (h["i"].astype(np.uint64) * np.uint64(step) & np.uint64(0xFFFFFFFF)).astype(np.float64)
        want_u = np.where(want < 0, want.astype(np.int64).astype(np.uint64), np.abs(want).astype(np.uint64))
        assert np.array_equal(h["freq_hz"], want_u)


@pytest.mark.parametrize("tag", ["12m5", "20m"])
def test_known_answer_tone_at_12m5_and_20_msps(torch_cuda, tag):
    g = np.load(os.path.join(GOLD, "known_answer_rates.npz"))
    n, fs, fc, m = (int(g[f"{tag}_n"]), int(g[f"{tag}_sample_rate"]), float(g[f"{tag}_center_freq"]), int(g[f"{tag}_tone_bin"]))
    x = (float(g[f"{tag}_amplitude"]) * np.exp(2j * np.pi * m * np.arange(n) / n)).astype(np.complex64)[None]
    for flags in (capi.OUT_SPECTRUM | capi.OUT_HITS, capi.OUT_HITS):
        with Plan(n, fs, float(g[f"{tag}_threshold"]), max_batch=1, flags=flags) as plan:
            plan.submit_device(0, _dev(torch_cuda, x), 1, [fc], [3])
            _, h, t = plan.collect(0, want_power=False)
        assert h["i"].tolist() == g[f"{tag}_hit_i"].tolist() and h["freq_hz"].tolist() == g[f"{tag}_hit_freq"].tolist()
        assert np.abs(h["power_db"] - g[f"{tag}_hit_db64"]).max() < 1e-5 and t.tolist() == [0]

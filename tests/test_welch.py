"""BASELINE config C5: streaming 65 536-pt, 50 %-overlap Welch PSD (no reference counterpart;
definition in SURVEY.md 8d).  CPU: the oracle against float64.  GPU: the HIP four-step path
against the oracle and float64, device-resident and pinned/hipGraph submits, double buffering."""
import numpy as np
import pytest

from scanner_amd import capi
from tests import tolerances as tol

N, K = 65536, 16


def _stream(n_psd, seed, k=K):
    rng = np.random.default_rng(seed)
    m = (n_psd * k + 1) * (N // 2)
    x = (rng.standard_normal(m) + 1j * rng.standard_normal(m)).astype(np.complex64) * np.float32(0.05)
    t = np.arange(m)
    for f, a in ((0.1234567, 0.3), (-0.31, 0.02), (0.25 + 3.3 / N, 0.1)):     # on- and off-bin tones
        x += (a * np.exp(2j * np.pi * f * t)).astype(np.complex64)
    return x


def test_oracle_welch_vs_float64(oracle_mod):
    x = _stream(1, seed=5, k=4)
    got = oracle_mod.welch(x, N, 4, 1)
    ref = oracle_mod.ref64_welch(x, oracle_mod.Oracle(N).window(), N, 4, 1)
    fig = tol.compare_spectra(got, ref)
    assert fig["max_rel_power_vs_max_bin_mean"] < 5e-6
    # a single segment (k=1) is exactly the single-FFT path of the scanner
    one = oracle_mod.welch(x, N, 1, 1)
    p, _, _ = oracle_mod.Oracle(N, threshold=1e9).run(x[:N])
    assert np.array_equal(one[0], p[0])


@pytest.mark.gpu
def test_welch_gpu_vs_oracle_and_float64(oracle_mod, built_lib):
    import torch

    from scanner_amd import WelchPlan

    assert torch.cuda.is_available()
    n_psd = 3
    x = _stream(n_psd, seed=5)
    ref = oracle_mod.welch(x, N, K, n_psd)
    ref64 = oracle_mod.ref64_welch(x, oracle_mod.Oracle(N).window(), N, K, n_psd)
    with WelchPlan(N, K, max_psd=4) as w:
        assert w.samples(n_psd) == (n_psd * K + 1) * (N // 2) == x.size
        d = torch.from_numpy(x.view(np.float32)).cuda()
        w.submit_device(0, d, n_psd)
        got = w.collect(0)
        print("welch HIP vs oracle :", tol.compare_spectra(got, ref))
        print("welch HIP vs float64:", tol.compare_spectra(got, ref64))
        # pinned + hipGraph path, both slots in flight, replayed twice (graph re-use), then a
        # different batch size (graph re-capture)
        xs = [_stream(2, seed=11 + s) for s in range(2)]
        for rep in range(2):
            for s in range(2):
                hb = w.host_buffer(s)
                hb[: xs[s].size] = xs[s]
                w.submit(s, 2)
            for s in range(2):
                g = w.collect(s)
                tol.compare_spectra(g, oracle_mod.welch(xs[s], N, K, 2))
        hb = w.host_buffer(0)
        hb[: x.size] = x
        w.submit(0, n_psd)
        g3 = w.collect(0)
        assert np.array_equal(g3, got)                      # same bits as the device-resident submit
        with pytest.raises(capi.ScannerError):
            w.submit(0, 5)                                  # > max_psd
        with pytest.raises(capi.ScannerError):
            w.collect(1)                                    # nothing pending
    # the tones sit where they should: strongest bin of the 0.25 + 3.3/N tone
    assert abs(int(np.argmax(got[0][N // 4 - 8: N // 4 + 8])) + N // 4 - 8 - (N // 4 + 3)) <= 1


@pytest.mark.gpu
def test_welch_k1_equals_single_fft_definition(oracle_mod, built_lib):
    import torch

    from scanner_amd import WelchPlan

    x = _stream(2, seed=9, k=1)
    with WelchPlan(N, 1, max_psd=2) as w:
        w.submit_device(0, torch.from_numpy(x.view(np.float32)).cuda(), 2)
        got = w.collect(0)
    tol.compare_spectra(got, oracle_mod.welch(x, N, 1, 2))
    with pytest.raises(capi.ScannerError):
        WelchPlan(4096, 4)                                  # only the 65 536-pt four-step exists

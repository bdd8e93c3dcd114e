"""CPU tests of the oracle (the float32 C restatement) against the committed goldens,
float64 mathematics and hand-derived known answers.  No GPU, no HIP calls."""
import os

import numpy as np
import pytest

from scanner_amd import capi, synth
from tests import tolerances as tol

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_window_matches_float64_and_scipy(oracle_mod):
    from scipy.signal.windows import blackmanharris

    for n in (64, 1024, 4096, 8192):
        w = oracle_mod.Oracle(n).window()
        assert w.dtype == np.float32
        assert np.abs(w - oracle_mod.ref64_window(n)).max() <= 6e-8        # float rounding only
        assert np.abs(w - blackmanharris(n, sym=True)).max() <= 1.2e-7      # [3P] same window


@pytest.mark.parametrize("n", [2, 8, 64, 1024, 4096, 8192, 65536])
def test_fft_matches_float64_dft(oracle_mod, n):
    rng = np.random.default_rng(n)
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    X = oracle_mod.Oracle(n).fft(x)
    X64 = np.fft.fft(x.astype(np.complex128))
    rms = np.sqrt((np.abs(X64) ** 2).mean())
    assert np.abs(X - X64).max() / rms < 2e-6
    # unnormalised, sign -1 (fft.cpp:10 FFTW_FORWARD): a unit impulse at n=1 gives exp(-2 pi i k/N)
    e = np.zeros(n, np.complex64)
    e[1] = 1
    E = oracle_mod.Oracle(n).fft(e)
    k = np.arange(n)
    assert np.abs(E - np.exp(-2j * np.pi * k / n)).max() < 1e-6


def test_magnitude_formula(oracle_mod):
    n = 1024
    rng = np.random.default_rng(5)
    X = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64) * 30
    X[7] = 0
    o = oracle_mod.Oracle(n)
    m = o.magnitude(X)
    ref = 5.0 * np.log10(X.real.astype(np.float64) ** 2 + X.imag.astype(np.float64) ** 2, where=np.abs(X) > 0,
                         out=np.full(n, -np.inf))
    assert m[7] == -np.inf                       # zero bin -> -inf, never a hit
    ok = np.isfinite(ref)
    assert np.abs(m[ok] - ref[ok]).max() < 2e-6
    # the log2f flavour differs by a few 1e-6 dB at most
    assert np.abs(o.magnitude(X, use_log2f=True)[ok] - m[ok]).max() < 5e-6


def test_known_answer_tone(oracle_mod):
    g = np.load(os.path.join(GOLD, "known_answer_tone.npz"))
    n = int(g["n"])
    x = (0.5 * np.exp(2j * np.pi * 128 * np.arange(n) / n)).astype(np.complex64)
    o = oracle_mod.Oracle(n, int(g["sample_rate"]), float(g["threshold"]))
    power, hits, trig = o.run(x, center_freqs=[float(g["center_freq"])], seq_ids=[7])
    assert hits["i"].tolist() == g["hit_i"].tolist()
    assert hits["freq_hz"].tolist() == g["hit_freq"].tolist()
    assert np.all(hits["seq_id"] == 7)
    assert np.abs(hits["power_db"] - g["hit_db64"]).max() < 5e-6
    # the transcript of SURVEY.md 8c, as process.cpp:57 would print it
    lines = ["freq %d power_db %f" % (h["freq_hz"], h["power_db"]) for h in hits]
    assert lines[2] == "freq 100999680 power_db 22.636375"
    assert trig.tolist() == [0]


@pytest.mark.parametrize("tag", ["12m5", "20m"])
def test_known_answer_tone_other_sample_rates(oracle_mod, tag):
    """process.cpp:38-39 where fs / N does not divide: bin_step = uint32(fs / N) truncated (12.5e6 / 8192 -> 1525,
    20e6 / 4096 -> 4882), start = fc - fs / 2 -- hand-derived in tests/golden/make_golden.py section 2b."""
    g = np.load(os.path.join(GOLD, "known_answer_rates.npz"))
    n, fs, fc, m = (int(g[f"{tag}_n"]), int(g[f"{tag}_sample_rate"]), float(g[f"{tag}_center_freq"]), int(g[f"{tag}_tone_bin"]))
    x = (float(g[f"{tag}_amplitude"]) * np.exp(2j * np.pi * m * np.arange(n) / n)).astype(np.complex64)
    _, hits, trig = oracle_mod.Oracle(n, fs, float(g[f"{tag}_threshold"])).run(x, center_freqs=[fc], seq_ids=[3])
    assert hits["i"].tolist() == g[f"{tag}_hit_i"].tolist()
    assert hits["freq_hz"].tolist() == g[f"{tag}_hit_freq"].tolist()
    assert np.abs(hits["power_db"] - g[f"{tag}_hit_db64"]).max() < 5e-6
    peak = len(hits) // 2
    assert int(hits["freq_hz"][peak]) == {"12m5": 435441400, "20m": 2408580936}[tag]      # 427670000 + 5096 * 1525; 2402000000 + 1348 * 4882
    assert trig.tolist() == [0]


def test_convert_known_answers(oracle_mod):
    k = np.load(os.path.join(GOLD, "convert_known_answers.npz"))
    O = oracle_mod

    def run(kind, enob, dc, arr):
        o = O.Oracle(arr.size // 2, kind=kind, enob=enob, correct_dc=dc)
        return o.convert(arr).view(np.float32).reshape(-1, 2)

    assert np.array_equal(run(O.KIND_SHORT_COMPLEX, 12, False, k["s16_enob12_in"]), k["s16_enob12_out"])
    assert np.array_equal(run(O.KIND_SHORT_COMPLEX, 16, False, k["s16_enob16_in"]), k["s16_enob16_out"])
    assert np.array_equal(run(O.KIND_SHORT_COMPLEX, 12, True, k["s16_dcneg_in"]), k["s16_dcneg_out"])
    assert np.array_equal(run(O.KIND_SHORT_COMPLEX, 12, True, k["s16_dcpos_in"]), k["s16_dcpos_out"])
    assert np.array_equal(run(O.KIND_BYTE_COMPLEX, 8, False, k["s8_enob8_in"]), k["s8_enob8_out"])
    # planar == interleaved on the same numbers (utility.cpp:9-32 vs :58-84)
    a = k["s16_dcpos_in"]
    planar = np.ascontiguousarray(a.T)
    assert np.array_equal(run(O.KIND_SHORT, 12, True, planar), k["s16_dcpos_out"])


@pytest.mark.parametrize("n", [1024, 4096, 8192, 16384])
def test_oracle_spectrum_vs_float64_golden(oracle_mod, n):
    g = np.load(os.path.join(GOLD, f"spectrum_n{n}.npz"))
    x = synth.cfloat_batch(n, int(g["n_buffers"]), int(g["seed"]))
    assert np.frombuffer(x.tobytes(), np.uint32).sum(dtype=np.uint64) == g["x_checksum"], "input recipe drifted"
    o = oracle_mod.Oracle(n, threshold=1e9)
    assert np.array_equal(o.window(), g["window_f32"])
    power, hits, trig = o.run(x)
    fig = tol.compare_spectra(power, g["db64"])
    assert fig["max_rel_power_vs_max_bin_mean"] < 5e-6
    assert len(hits) == 0 and not trig.any()


def _brute_hits(db, n, fs, fc, thr, use_bw=0.75, dcw=4, trig_count=1047):
    """process.cpp:36-64 as a plain Python loop."""
    half, use_window = n // 2, int(use_bw * n / 2.0)
    start = fc - fs // 2
    step = fs // n
    out = []
    for i in range(n):
        j = (i + half) % n
        if j < dcw or (n - j) < dcw:
            continue
        if i < half - use_window or i > half + use_window:
            continue
        if db[j] > np.float32(thr):
            out.append((i, int(start + ((i * step) & 0xFFFFFFFF))))
    return out, len(out) > trig_count


def test_process_fft_mask_and_trigger(oracle_mod):
    n, fs, fc = 4096, 8000000, 433.5e6
    x = synth.cfloat_batch(n, 3, seed=9, sigma=0.1)
    o = oracle_mod.Oracle(n, fs, threshold=-100.0)          # everything evaluated is a hit
    power, hits, trig = o.run(x, center_freqs=[fc] * 3)
    m = tol.evaluated_mask(n)
    assert m.sum() == 2 * 1536 + 1 - 7                       # 3066 of 4096
    assert len(hits) == 3 * m.sum() and trig.tolist() == [1, 1, 1]
    for b in range(3):
        bh, bt = _brute_hits(power[b], n, fs, fc, -100.0)
        mine = hits[hits["seq_id"] == b]
        assert [(int(h["i"]), int(h["freq_hz"])) for h in mine] == bh
        assert bt
    # strict '>' (process.cpp:54): a threshold equal to a bin's value does not report it
    peak = power[0][m].max()
    o2 = oracle_mod.Oracle(n, fs, threshold=float(peak))
    _, h2, _ = o2.run(x[:1], center_freqs=[fc])
    assert len(h2) == 0
    # hits come sorted by (buffer, i)
    key = hits["seq_id"].astype(np.int64) * n + hits["i"]
    assert np.all(np.diff(key) > 0)


def test_run_batch_threads_equal_single(oracle_mod):
    n = 1024
    x = synth.cfloat_batch(n, 37, seed=3, sigma=0.1)
    o = oracle_mod.Oracle(n, threshold=8.0)
    fc = 100e6 + 6e6 * np.arange(37)
    p1, h1, t1 = o.run(x, fc, threads=1)
    p8, h8, t8 = o.run(x, fc, threads=8)
    assert np.array_equal(p1, p8) and np.array_equal(h1, h8) and np.array_equal(t1, t8)
    assert len(h1) > 0


def test_int_kinds_through_whole_path(oracle_mod):
    O = oracle_mod
    n = 1024
    x = synth.cfloat_batch(n, 2, seed=11, sigma=0.1)
    raw16 = synth.quantize(x, capi.KIND_SHORT_COMPLEX)
    p_i, _, _ = O.Oracle(n, kind=O.KIND_SHORT_COMPLEX, enob=12, threshold=1e9).run(raw16)
    p_p, _, _ = O.Oracle(n, kind=O.KIND_SHORT, enob=12, threshold=1e9).run(synth.quantize(x, capi.KIND_SHORT))
    assert np.array_equal(p_i, p_p)
    # against float64 on the exactly-converted samples
    conv = raw16.astype(np.float32) / np.float32(2048)
    xc = (conv[..., 0] + 1j * conv[..., 1]).astype(np.complex64)
    _, _, db64 = O.ref64_spectrum(xc, O.Oracle(n).window())
    tol.compare_spectra(p_i, db64)
    # int8 with enob 8: sign-flipped samples (utility.cpp:40), identical power spectrum
    raw8 = synth.quantize(x, capi.KIND_BYTE_COMPLEX)
    p8, _, _ = O.Oracle(n, kind=O.KIND_BYTE_COMPLEX, enob=8, threshold=1e9).run(raw8)
    c8 = raw8.astype(np.float32) / np.float32(-128)
    _, _, db8 = O.ref64_spectrum((c8[..., 0] + 1j * c8[..., 1]).astype(np.complex64), O.Oracle(n).window())
    tol.compare_spectra(p8, db8)


def test_time_domain(oracle_mod):
    n = 1024
    x = np.full(n, 0.001 + 0j, np.complex64)
    x[100] = 10.0
    hit, mx, mn = oracle_mod.Oracle(n).time_domain(x, threshold=9.0)
    assert hit and abs(mx - 10.0) < 1e-5 and abs(mn + 30.0) < 1e-4
    # all magnitudes below 0 dB: max keeps its odd initial value (process.cpp:207)
    hit, mx, _ = oracle_mod.Oracle(n).time_domain(np.full(n, 0.5 + 0j, np.complex64), threshold=0.0)
    assert hit and mx == np.float32(1.17549435e-38)


def test_frequency_table(oracle_mod):
    t = oracle_mod.frequency_table(8000000, 0.0, 16384 * 6e6)
    assert len(t) == 16384 and t[0] == 3e6 and t[1] - t[0] == 6e6
    assert len(oracle_mod.frequency_table(8000000, 88e6, 0.0)) == 1    # stop == 0 -> single centre
    t2 = oracle_mod.frequency_table(8000000, 88e6, 108e6)
    assert np.all(t2 < 108e6) and t2[-1] + 6e6 >= 108e6
    assert len(t2) == int(np.ceil((108e6 - t2[0]) / 6e6))              # the assert of frequencyTable.cpp:29


def test_fft_modes_bracket_the_float64_transform(oracle_mod):
    """The oracle's default FFT arithmetic (double-internal, rounded once: the FFTW-accuracy stand-in) and its
    textbook float radix-2 mode (what bench.py times) are two evaluations of the same DFT: both within the parity
    bar of the float64 transform, the default one essentially exact."""
    import numpy as np
    from scanner_amd import synth
    from tests import tolerances as tol

    n, nb = 8192, 24
    x = synth.cfloat_batch(n, nb, seed=77)
    o = oracle_mod.Oracle(n, 8000000, 10.0)
    _, P64, _ = oracle_mod.ref64_spectrum(x, o.window())
    mean = P64.mean(axis=-1, keepdims=True)
    errs = {}
    try:
        for accurate in (True, False):
            oracle_mod.set_fft_mode(accurate)
            p, _, _ = o.run(x, want_hits=False)
            errs[accurate] = float((np.abs(tol.db_to_power(p) - P64) / np.maximum(P64, mean)).max())
    finally:
        oracle_mod.set_fft_mode(True)
    assert errs[True] < 1e-6 and errs[False] < tol.REL_POWER, errs
    assert errs[True] < errs[False]


def test_every_gnuradio_window_type(oracle_mod):
    """process.cpp:18 hands ANY gr::fft::window::win_type to window::build(type, N, 0.0) ([3P], GNU Radio 3.7 / 3.8): the oracle
    builds all eight from the published definitions -- held here to an independent float64 evaluation and, where scipy uses the
    same coefficients, to scipy.signal.windows (symmetric form)."""
    import scipy.signal.windows as W

    O = oracle_mod
    for n in (16, 1000, 4096):
        for t in range(8):
            w = O.window(t, n)
            assert w.dtype == np.float32 and np.abs(w - O.ref64_window_of(t, n)).max() < 6e-8, (t, n)
        for t, f in ((O.WIN_HAMMING, W.hamming), (O.WIN_HANN, W.hann), (O.WIN_BLACKMAN, W.blackman), (O.WIN_BARTLETT, W.bartlett),
                     (O.WIN_BLACKMAN_HARRIS, W.blackmanharris)):
            assert np.abs(O.window(t, n) - f(n, sym=True)).max() < 1.2e-7, (t, n)
        assert np.all(O.window(O.WIN_RECTANGULAR, n) == 1) and np.all(O.window(O.WIN_KAISER, n) == 1)   # Kaiser with beta = 0.0
        assert np.array_equal(O.window(O.WIN_BLACKMAN_HARRIS, n), O.Oracle(n).window())
        ft = O.window(O.WIN_FLATTOP, n).astype(np.float64)
        assert (n < 1000 or abs(ft.max() - (1 + 1.93 + 1.29 + 0.388 + 0.028) / 4.63867) < 1e-3) and ft.min() < 0   # its peak (between two samples of an even length) and its negative lobes
    # run_batch follows the switch, and the switch resets
    x = (np.random.default_rng(3).standard_normal((2, 1024, 2)) * 0.1).astype(np.float32).view(np.complex64).reshape(2, 1024)
    with O.window_type(O.WIN_HANN):
        p = O.Oracle(1024, threshold=1e9).run(x)[0]
    _, _, db64 = O.ref64_spectrum(x, O.window(O.WIN_HANN, 1024))
    from tests import tolerances as tol
    tol.compare_spectra(p, db64)
    assert np.array_equal(O.Oracle(1024, threshold=1e9).run(x)[0], O.Oracle(1024, threshold=1e9).run(x)[0])
    _, _, db_bh = O.ref64_spectrum(x, O.Oracle(1024).window())
    tol.compare_spectra(O.Oracle(1024, threshold=1e9).run(x)[0], db_bh)

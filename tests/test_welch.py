"""BASELINE config C5: streaming 65 536-pt, 50 %-overlap Welch PSD (no reference counterpart;
definition in SURVEY.md 8d).  CPU: the oracle against float64.  GPU: the HIP four-step path
against the oracle and float64 -- at the shape bench.py times (32 PSDs per submit: 11 segments per column
workgroup with the in-register carry of the overlapping half, one row workgroup per PSD row tile over all 16
segments) and at the edges of its partitioning (ragged column groups, 1 / 2 / 4 row parts with K not a multiple),
device-resident and pinned/hipGraph submits, all slots in flight, every wire format."""
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


def _nonstationary(x, k, seed):
    """Every segment gets its own level (a gain step per delivery block): a PSD that takes a wrong segment, or the wrong
    half of one, differs by whole dB -- which a stationary stream would hide."""
    rng = np.random.default_rng(seed)
    g = rng.uniform(0.2, 1.0, x.size // (N // 2)).astype(np.float32)
    return (x.reshape(-1, N // 2) * g[:, None]).reshape(-1)


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


def test_oracle_welch_raw_is_convert_per_block_then_welch(oracle_mod):
    """The wire-format definition: K1 per delivery block of N/2 samples (with DC removal the block's own integer mean),
    then the cfloat definition.  Pinned against the oracle's converters, which are held to the reference's utility.cpp."""
    from scanner_amd import synth

    k, n_psd, hop = 2, 1, N // 2
    x = _stream(n_psd, seed=3, k=k) + np.complex64(0.11 + 0.07j)            # a DC offset for correct_dc to remove (positive sums:
    # a negative sum meets the int32 /= uint32 quirk of utility.cpp:77-78, exercised on the GPU below)
    for kind, enob in ((capi.KIND_SHORT_COMPLEX, 12), (capi.KIND_SHORT, 12), (capi.KIND_BYTE_COMPLEX, 8)):
        raw = synth.quantize(x.reshape(-1, hop), kind)
        for dc in (False, True):
            conv = oracle_mod.welch_convert(raw, kind, enob, dc, hop)
            assert conv.size == x.size
            blocks = [oracle_mod.Oracle(hop, kind=kind, enob=enob, correct_dc=dc).convert(raw[b]) for b in range(raw.shape[0])]
            assert np.array_equal(conv, np.concatenate(blocks))
            got = oracle_mod.welch_raw(raw, kind, enob, dc, N, k, n_psd)
            assert np.array_equal(got, oracle_mod.welch(conv, N, k, n_psd))
        # removing the mean moves the DC bin by tens of dB and nothing else by more than the leakage of the block steps
        a = oracle_mod.welch_raw(raw, kind, enob, False, N, k, n_psd)[0]
        b = oracle_mod.welch_raw(raw, kind, enob, True, N, k, n_psd)[0]
        assert a[0] - b[0] > 10.0


def _check(got, x_c64, k, n_psd, oracle_mod, label):
    ref = oracle_mod.welch(x_c64, N, k, n_psd)
    ref64 = oracle_mod.ref64_welch(x_c64, oracle_mod.Oracle(N).window(), N, k, n_psd)
    f1 = tol.compare_spectra(got, ref)
    f2 = tol.compare_spectra(got, ref64)
    print(f"welch {label}: vs oracle {f1['max_rel_power_vs_max_bin_mean']:.2e}, vs float64 {f2['max_rel_power_vs_max_bin_mean']:.2e}")
    return ref


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


@pytest.mark.gpu
def test_welch_at_the_bench_shape_and_column_group_edges(oracle_mod, built_lib):
    """WelchPlan(65536, 16, max_psd=32) -- bench.py's C5 plan: one row workgroup per PSD row tile accumulates all 16
    segments (parts == 1), and with 256 CUs the column kernel runs 48 groups of consecutive segments per tile:
      n_psd  4 ->  64 segments: 2 per group, 32 groups used           (the carry consumed once)
      n_psd  5 ->  80 segments: 2 per group, 40 groups
      n_psd 13 -> 208 segments: 5 per group, last group ragged (3)
      n_psd 32 -> 512 segments: 11 per group, last group ragged (6)   (the shape the bench times)
    The stream is non-stationary (a gain step per delivery block), so a segment assembled from the wrong carried half shows."""
    import torch

    from scanner_amd import WelchPlan

    cus = torch.cuda.get_device_properties(0).multi_processor_count
    with WelchPlan(N, K, max_psd=32) as w:
        for n_psd in (4, 5, 13, 32):
            parts, groups, per = w.partition(n_psd)
            assert parts == 1
            if cus == 256:
                assert (groups, per) == (48, {4: 2, 5: 2, 13: 5, 32: 11}[n_psd])
            assert per >= 2                                                       # the in-register carry is consumed
            x = _nonstationary(_stream(n_psd, seed=100 + n_psd), K, seed=n_psd)
            d = torch.from_numpy(x.view(np.float32)).cuda()
            w.submit_device(0, d, n_psd)
            got = w.collect(0)
            ref = _check(got, x, K, n_psd, oracle_mod, f"device n_psd={n_psd} groups={groups} per={per}")
            # the non-stationary stream does tell segments apart: PSD p against PSD p+1 is off by far more than the bar
            assert np.abs(ref[0] - ref[1]).max() > 0.5
            # the same submit through pinned memory and the captured graph, slots 1 and 2 in flight together
            for s in (1, 2):
                hb = w.host_buffer(s)
                hb[: x.size] = x if s == 1 else x[::-1]
                w.submit(s, n_psd)
            g1, g2 = w.collect(1), w.collect(2)
            assert np.array_equal(g1, got)
            tol.compare_spectra(g2, oracle_mod.welch(np.ascontiguousarray(x[::-1]), N, K, n_psd))


@pytest.mark.gpu
@pytest.mark.parametrize("k,max_psd,n_psds,want_parts", [
    (2, 16, (16, 7), 1),      # parts == 1 (16 tiles x 16 PSDs fill the CUs), K = 2
    (3, 16, (16, 5), 1),      # parts == 1, K = 3
    (16, 16, (16, 3), 1),     # parts == 1, K = 16, few PSDs: column groups of 1 segment .. 6
    (3, 8, (8, 3), 2),        # parts == 2, K = 3: parts of 1 and 2 segments
    (7, 8, (8, 1), 2),        # parts == 2, K = 7: 3 + 4
    (5, 2, (2, 1), 4),        # parts == 4, K = 5: 1 + 1 + 1 + 2
    (7, 4, (4, 3), 4),        # parts == 4, K = 7: 1 + 2 + 2 + 2
    (16, 2, (2,), 4),         # parts == 4, K = 16 (the r02 8-PSD shape's split)
])
def test_welch_row_parts_and_segment_counts(oracle_mod, built_lib, k, max_psd, n_psds, want_parts):
    import torch

    from scanner_amd import WelchPlan

    cus = torch.cuda.get_device_properties(0).multi_processor_count
    with WelchPlan(N, k, max_psd=max_psd) as w:
        for n_psd in n_psds:
            parts, groups, per = w.partition(n_psd)
            if cus == 256:
                assert parts == want_parts, (parts, want_parts)
            x = _nonstationary(_stream(n_psd, seed=7 * k + n_psd, k=k), k, seed=k)
            w.submit_device(0, torch.from_numpy(x.view(np.float32)).cuda(), n_psd)
            got = w.collect(0)
            _check(got, x, k, n_psd, oracle_mod, f"k={k} max_psd={max_psd} n_psd={n_psd} parts={parts} groups={groups} per={per}")
            hb = w.host_buffer(3)
            hb[: x.size] = x
            w.submit(3, n_psd)
            assert np.array_equal(w.collect(3), got)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,enob", [(capi.KIND_SHORT_COMPLEX, 12), (capi.KIND_SHORT, 12), (capi.KIND_BYTE_COMPLEX, 8),
                                       (capi.KIND_SHORT_COMPLEX, 16)])
@pytest.mark.parametrize("dc", [False, True])
def test_welch_wire_formats(oracle_mod, built_lib, kind, enob, dc):
    """K1 inside the Welch path (utility.cpp:9-84 per delivery block of N/2 samples): int16 interleaved, int16 planar, int8,
    with and without DC removal, K = 16, at a shape where column workgroups carry halves across segments (and, with DC
    removal, across blocks of different means).  One block has a NEGATIVE integer sum: the int32 /= uint32 quirk of
    utility.cpp:77-78 turns its mean into a huge positive number, which the GPU must reproduce."""
    import torch

    from scanner_amd import WelchPlan, synth

    n_psd, hop = 4, N // 2
    x = _nonstationary(_stream(n_psd, seed=31 + kind), K, seed=kind) * np.float32(0.5)
    x = x.reshape(-1, hop).copy()
    x += np.complex64(0.02 + 0.01j)
    x[5] -= np.complex64(0.05 + 0.05j)                       # block 5: negative sums
    raw = synth.quantize(x, kind)
    ref = oracle_mod.welch_raw(raw, kind, enob, dc, N, K, n_psd)
    conv = oracle_mod.welch_convert(raw, kind, enob, dc, hop)
    if dc:
        assert np.abs(conv[5 * hop:6 * hop]).max() > 3.0     # the quirk is in play (2^32 / 32768 = 131072 counts: -64 at enob 12, +4 at 16, +1024 at int8)
    with WelchPlan(N, K, max_psd=16, kind=kind, enob=enob, correct_dc=dc) as w:
        assert w.partition(n_psd)[2] >= 2
        flat = np.ascontiguousarray(raw).view(np.uint8).reshape(-1)
        assert flat.size == w.samples(n_psd) * w.bytes_per_sample
        w.submit_device(0, torch.from_numpy(flat).cuda(), n_psd)
        got = w.collect(0)
        print("welch wire format vs oracle :", tol.compare_spectra(got, ref))
        tol.compare_spectra(got, oracle_mod.ref64_welch(conv, oracle_mod.Oracle(N).window(), N, K, n_psd))
        hb = w.host_buffer(1)
        assert hb.dtype == np.uint8
        hb[: flat.size] = flat
        w.submit(1, n_psd)
        assert np.array_equal(w.collect(1), got)
    with pytest.raises(capi.ScannerError):
        WelchPlan(N, K, max_psd=1, kind=capi.KIND_BYTE_COMPLEX, enob=9)

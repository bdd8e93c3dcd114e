"""Every shipped kernel specialisation of the fused path, deterministically: the sizes the LIBRARY says it runs fused
(scn_size_path, so this list cannot drift from scn_api.hip's dispatch) x the four wire formats x DC removal off / on x the
three output modes (spectrum only, spectrum + hits, hits only), each against the oracle on a seeded batch that is large
enough to take every workgroup of the persistent launch through more than one buffer (prefetch, buffer queue, ragged last
iteration) and whose threshold -- guard-band-free on the oracle's spectra -- sits in the noise tail, so that thousands of
noise bins and every tone are reported.

Asserted per (size, format, DC):
  * spectrum + hits: spectra to the parity bar; the hit list (seq_id, i, freq_hz, order) and the trigger flags bit for bit
    (process.cpp:54 `magnitudes[j] > m_threshold`, :62); every record's power_db IS the float the spectrum holds;
  * hits only (the kernels that take the decision on linear power first and have no stores -- what the in-tree
    ProcessSamples worker runs): byte-identical records to the spectrum + hits plan, power_db included;
  * spectrum only: byte-identical spectra to the spectrum + hits plan."""
import functools

import numpy as np
import pytest

from scanner_amd import Plan, build, capi, synth
from tests import tolerances as tol

pytestmark = pytest.mark.gpu
FS = 8000000

build.build()  # (collection needs scn_size_path; seconds when the library is current, and it needs no GPU)
FUSED_SIZES = [1 << k for k in range(4, 17) if capi.size_path(1 << k) == capi.PATH_FUSED]
# the mixed-radix fused kernels (scn_mixed.hip): every size the library runs fused that is not a power of two
MIXED_SIZES = [n for n in range(17, 16384) if n & (n - 1) and capi.size_path(n) == capi.PATH_FUSED]
FORMATS = [(capi.KIND_FLOAT_COMPLEX, 12, False)] + [(k, e, dc) for k, e in ((capi.KIND_SHORT_COMPLEX, 12), (capi.KIND_SHORT, 14),
                                                                             (capi.KIND_BYTE_COMPLEX, 8)) for dc in (False, True)]
NAMES = {capi.KIND_FLOAT_COMPLEX: "cfloat", capi.KIND_SHORT_COMPLEX: "int16", capi.KIND_SHORT: "int16planar", capi.KIND_BYTE_COMPLEX: "int8"}


def test_the_fused_sizes_are_the_documented_ones():
    assert FUSED_SIZES == [16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384]
    assert capi.size_path(32768) == capi.size_path(65536) == capi.PATH_FOUR_STEP
    assert MIXED_SIZES == [1000, 1200, 1280, 1500, 1536, 1920, 2000, 2400, 2500, 2560, 3000, 3072, 3600, 3840, 4000, 4800, 5000, 5120, 6000,
                           6144, 7200, 7680, 8000, 9000, 9600, 10000,
                           10240, 12000, 12288, 12800, 14400, 15000, 15360, 16000]   # (from 10240 up: the two-virtual-thread, in-place form)
    assert capi.size_path(1023) == capi.size_path(17) == capi.size_path(11000) == capi.size_path(65535) == capi.PATH_BLUESTEIN
    assert capi.size_path(8) == capi.size_path(65537) == capi.PATH_UNSUPPORTED


@functools.lru_cache(maxsize=2)
def _batch(n):
    nb = 5500000 // n + 3   # 21487 x 256 ... 338 x 16384: more buffers than a launch has resident workgroup slots, ragged
    return synth.cfloat_batch(n, nb, seed=4000 + n, sigma=0.05)


def _to_dev(torch, raw):
    return torch.from_numpy(np.ascontiguousarray(raw).view(np.uint8).reshape(-1)).cuda()


def _drop_guard_band(h, near, seq0, n):
    """the records whose bin is NOT within the guard band of the threshold on the oracle's spectrum"""
    b = (h["seq_id"] - seq0).astype(np.int64)
    j = (h["i"].astype(np.int64) + n // 2) % n
    return h[~near[b, j]]


# the mixed-radix sizes: every kernel of a size is (wire format) x (output mode) -- DC removal is a runtime branch there --, so four
# formats per size reach all twelve (DC on for two of them); two sizes walk all seven combinations like the powers of two
MIXED_FORMATS = [(capi.KIND_FLOAT_COMPLEX, 12, False), (capi.KIND_SHORT_COMPLEX, 12, True), (capi.KIND_SHORT, 14, False), (capi.KIND_BYTE_COMPLEX, 8, True)]
CASES = [(n, *f) for n in FUSED_SIZES for f in FORMATS] + [(n, *f) for n in MIXED_SIZES for f in (FORMATS if n in (1000, 6000, 12000) else MIXED_FORMATS)]


@pytest.mark.parametrize("n,kind,enob,dc", CASES, ids=[f"{n}-{NAMES[k]}{'-dc' if dc else ''}" for n, k, _, dc in CASES])
def test_every_fused_specialisation_vs_oracle(built_lib, oracle_mod, n, kind, enob, dc):
    import torch

    assert torch.cuda.is_available(), "-m gpu tests need a GPU; refusing to skip silently"
    x = _batch(n)
    nb = len(x)
    small = max(8, 262144 // n)   # the leading buffers are also launched on their own (a second slot, in flight beside the big one)
    raw = synth.quantize(x, kind)
    if dc:
        # a positive mean for the integer DC removal to take out.  The reference's `int32 /= uint32` turns a NEGATIVE sum into
        # a mean of ~2^32 / n (utility.cpp:77-78; pinned by test_dc_quirk_negative_mean): such a buffer is one huge constant
        # whose other bins are cancellation residue, where two correct float32 FFTs differ by whole dB -- so the few buffers
        # whose I or Q sum is not positive (a strong tone below one bin of frequency) are replaced by the first good one
        info = np.iinfo(raw.dtype)
        # (below 256 points a tone's sum over the buffer is no longer small against n times the offset: a larger offset, and
        #  more replaced buffers allowed)
        off = (60 if n >= 256 else 400) if raw.dtype == np.int16 else (11 if n >= 256 else 30)
        raw = np.clip(raw.astype(np.int32) + off, info.min, info.max).astype(raw.dtype)
        sums = raw.astype(np.int64).sum(axis=2 if kind == capi.KIND_SHORT else 1)   # [B, 2]: I and Q sums of every buffer
        bad = (sums <= 0).any(axis=1)
        assert bad.mean() < (0.05 if n >= 256 else 0.3)
        raw[bad] = raw[np.flatnonzero(~bad)[0]]
    fc = 70e6 + 6e6 * np.arange(nb)
    seq = np.arange(nb, dtype=np.uint64) + (1 << 33)
    p_ref, _, _ = oracle_mod.Oracle(n, FS, 1e9, kind=kind, enob=enob, correct_dc=dc).run(raw, want_hits=False, threads=8)
    ev = tol.evaluated_mask(n)
    # The threshold sits in the noise tail: a noise bin's power is exponentially distributed, P(p > k * mean) = exp(-k); mean =
    # median / ln 2 = median + 0.8 dB (5 log10 units), k = 6.9 (one bin in a thousand) = + 4.2 dB.  It is guard-band-free on
    # the SMALL launch's ~200 k evaluated bins, whose hit list is therefore demanded bit for bit; among the big launch's ~4 M
    # bins a handful sit within the guard band of ANY threshold that still reports noise, so there every record outside the
    # guard band is demanded, bit for bit and in order, and the rest is bounded by the band's population.
    thr = tol.pick_threshold(p_ref[:small], n, start=float(np.median(p_ref[:, ev])) + 5.0)
    o = oracle_mod.Oracle(n, FS, thr, kind=kind, enob=enob, correct_dc=dc)
    _, h_ref, _ = o.run(raw, fc, seq, want_power=False, threads=8)
    # process_fft's return value, hits > trigger_count (process.cpp:62), with the count at the batch's median so that about
    # half of the flags are set
    trig_count = max(1, int(np.median(np.bincount((h_ref["seq_id"] - seq[0]).astype(np.int64), minlength=nb))))  # (0 means "the reference's 1047" to a plan)
    o.params.trigger_count = trig_count
    _, h_ref, t_ref = o.run(raw, fc, seq, want_power=False, threads=8)
    hs_ref = h_ref[h_ref["seq_id"] < seq[0] + np.uint64(small)]
    assert len(h_ref) > 2000 and len(hs_ref) > 50 and t_ref.any() and not t_ref.all()
    near = np.zeros(p_ref.shape, bool)
    near[:, ev] = np.abs(p_ref[:, ev].astype(np.float64) - thr) < tol.GUARD_DB
    assert not near[:small].any()
    # a buffer's trigger flag is only in question if the guard band could move its count across trigger_count
    counts_ref = np.bincount((h_ref["seq_id"] - seq[0]).astype(np.int64), minlength=nb)
    trig_safe = (counts_ref + near.sum(axis=1) <= trig_count) | (counts_ref - near.sum(axis=1) > trig_count)
    d_raw = _to_dev(torch, raw)
    cap = len(h_ref) + int(near.sum()) + 4096
    got = {}
    for mode, fl in (("both", capi.OUT_SPECTRUM | capi.OUT_HITS), ("hits", capi.OUT_HITS), ("spectrum", capi.OUT_SPECTRUM)):
        with Plan(n, FS, thr, kind=kind, enob=enob, correct_dc=dc, max_batch=nb, max_hits=cap, flags=fl, trigger_count=trig_count) as plan:
            plan.submit_device(2, d_raw, nb, fc, seq)
            plan.submit_device(1, d_raw, small, fc[:small], seq[:small])
            got[mode] = (plan.collect(2, hit_cap=cap), plan.collect(1, hit_cap=cap))
    (p, h, t), (p1, h1, t1) = got["both"]
    fig = tol.compare_spectra(p, p_ref)
    assert p1.tobytes() == p[:small].tobytes()
    # the small launch: bit for bit
    assert len(h1) == len(hs_ref), (len(h1), len(hs_ref))
    for f in ("seq_id", "i", "freq_hz"):
        assert np.array_equal(h1[f], hs_ref[f]), f
    assert np.array_equal(t1, t_ref[:small])
    # the big launch: bit for bit outside the guard band
    g, r = _drop_guard_band(h, near, seq[0], n), _drop_guard_band(h_ref, near, seq[0], n)
    assert len(g) == len(r), (len(g), len(r))
    for f in ("seq_id", "i", "freq_hz"):
        assert np.array_equal(g[f], r[f]), f
    assert abs(len(h) - len(h_ref)) <= int(near.sum())
    assert np.array_equal(t[trig_safe], t_ref[trig_safe])
    for hh, pp, s0 in ((h, p, seq[0]), (h1, p1, seq[0])):
        jj = (hh["i"].astype(np.int64) + n // 2) % n
        assert np.array_equal(hh["power_db"], pp[(hh["seq_id"] - s0).astype(np.int64), jj]), "a record carries the float the spectrum holds"
    # hits-only plans (the decision taken on linear power first, no stores): byte-identical records
    (ph, hh, th), (ph1, hh1, th1) = got["hits"]
    assert ph is None and ph1 is None
    assert hh.tobytes() == h.tobytes() and hh1.tobytes() == h1.tobytes(), "hits-only and spectrum + hits plans report identical records"
    assert np.array_equal(th, t) and np.array_equal(th1, t1)
    # spectrum-only plans: byte-identical spectra
    (ps, hs, ts), (ps1, _, _) = got["spectrum"]
    assert hs is None and ts is None
    assert ps.tobytes() == p.tobytes() and ps1.tobytes() == p1.tobytes(), "spectrum-only and spectrum + hits plans store identical spectra"
    print(f"n={n} {NAMES[kind]} dc={dc}: {nb} + {small} buffers, thr {thr:.2f} dB, {len(h_ref)} + {len(hs_ref)} hits, {int(t_ref.sum())} triggers, "
          f"{int(near.sum())} bins in the guard band, max rel power {fig['max_rel_power_vs_max_bin_mean']:.2e}")

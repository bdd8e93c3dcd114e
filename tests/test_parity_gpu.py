"""Parity tests proper: the HIP path, called through the C-ABI, against the CPU oracle on
identical seeded inputs, against the committed float64 goldens, and -- at BASELINE.json's
full batch size -- against the oracle on every buffer plus size-independent properties.

Bar (tests/tolerances.py): power spectra within 1e-5 relative (to max(bin, buffer mean));
hit lists (bin index i, frequency, order, trigger flag) bit-exact."""
import os

import numpy as np
import pytest

from scanner_amd import Plan, capi, synth
from tests import tolerances as tol

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FS = 8000000


@pytest.fixture(scope="module")
def torch_cuda(built_lib):
    import torch

    assert torch.cuda.is_available(), "-m gpu tests need a GPU; refusing to skip silently"
    return torch


def _to_dev(torch, raw):
    return torch.from_numpy(np.ascontiguousarray(raw).view(np.uint8).reshape(-1)).cuda()


def _assert_hits_equal(got, ref):
    assert len(got) == len(ref), (len(got), len(ref))
    for f in ("seq_id", "i", "freq_hz"):
        assert np.array_equal(got[f], ref[f]), f
    # the reported value is the float the spectrum holds (spectra are compared bin by bin by
    # tol.compare_spectra; a detection can sit on a bin far below the buffer mean, where two float32
    # FFTs legitimately differ by more than the dB bar, so only a sanity bound here)
    assert np.abs(got["power_db"].astype(np.float64) - ref["power_db"]).max(initial=0) < 2e-2


def _assert_hits_equal_outside_guard(got, ref, p_ref, n, thr):
    """For batches too large for a guard-band-free threshold to exist: every record whose bin is NOT within tol.GUARD_DB of
    the threshold on the oracle's spectrum must be there, bit for bit and in order -- always asserted; returns the
    guard band's population (the caller bounds what may differ inside it)."""
    m = tol.evaluated_mask(n)
    near = np.zeros(p_ref.shape, bool)
    near[:, m] = np.abs(p_ref[:, m].astype(np.float64) - thr) < tol.GUARD_DB

    def outside(a, seq0):
        b = (a["seq_id"] - seq0).astype(np.int64)
        j = (a["i"].astype(np.int64) + n // 2) % n
        return a[~near[b, j]]

    seq0 = ref["seq_id"].min() if len(ref) else (got["seq_id"].min() if len(got) else 0)
    g, r = outside(got, seq0), outside(ref, seq0)
    assert len(g) == len(r), (len(g), len(r))
    for f in ("seq_id", "i", "freq_hz"):
        assert np.array_equal(g[f], r[f]), f
    assert abs(len(got) - len(ref)) <= int(near.sum())
    return int(near.sum())


def _clear_threshold(oracle_mod, n, raws, start, kind=capi.KIND_FLOAT_COMPLEX, enob=12, correct_dc=False):
    """A threshold >= start whose guard band (tol.GUARD_DB) holds no evaluated bin of ANY of the batches `raws` on the
    oracle's spectra, so that the hit lists can be demanded bit for bit -- unconditionally."""
    o = oracle_mod.Oracle(n, FS, 1e9, kind=kind, enob=enob, correct_dc=correct_dc)
    spectra = [o.run(r, want_hits=False, threads=8)[0] for r in raws if len(r)]
    return tol.pick_threshold(np.concatenate(spectra), n, start=start) if spectra else float(start)


def _run_both(torch, oracle_mod, n, kind, raw, fc, seq, thr, enob=12, correct_dc=False, slot=0, max_batch=None,
              max_hits=None):
    nb = len(fc)
    max_hits = max_hits or max(1024, nb * 256)
    o = oracle_mod.Oracle(n, FS, thr, kind=kind, enob=enob, correct_dc=correct_dc)
    p_ref, h_ref, t_ref = o.run(raw, fc, seq, threads=4)
    with Plan(n, FS, thr, kind=kind, enob=enob, correct_dc=correct_dc, max_batch=max_batch or nb,
              max_hits=max_hits) as plan:
        plan.submit_device(slot, _to_dev(torch, raw), nb, fc, seq)
        p, h, t = plan.collect(slot, hit_cap=max_hits)
    return (p, h, t), (p_ref, h_ref, t_ref)


# ---------------------------------------------------------------------------------------
# BASELINE config C2 shape (4096-pt cfloat), oracle-sized
# ---------------------------------------------------------------------------------------
def test_c2_cfloat_4096_vs_oracle_and_golden(torch_cuda, oracle_mod):
    n, nb = 4096, 96
    x = synth.cfloat_batch(n, nb, seed=2)
    fc = 3e6 + 6e6 * np.arange(nb)
    seq = np.arange(1000, 1000 + nb, dtype=np.uint64)
    o = oracle_mod.Oracle(n, FS, 1e9)
    p_ref, _, _ = o.run(x, threads=4)
    thr = tol.pick_threshold(p_ref, n, start=8.0)
    (p, h, t), (p_ref, h_ref, t_ref) = _run_both(torch_cuda, oracle_mod, n, capi.KIND_FLOAT_COMPLEX, x, fc, seq, thr)
    fig = tol.compare_spectra(p, p_ref)
    print("C2 vs oracle:", fig)
    assert len(h_ref) > 50, "test input should produce detections"
    _assert_hits_equal(h, h_ref)
    assert np.array_equal(t, t_ref)
    # both also sit on the float64 mathematics
    _, _, db64 = oracle_mod.ref64_spectrum(x, o.window())
    print("C2 vs float64:", tol.compare_spectra(p, db64))


def test_golden_4096(torch_cuda, oracle_mod):
    g = np.load(os.path.join(GOLD, "spectrum_n4096.npz"))
    n, nb = 4096, int(g["n_buffers"])
    x = synth.cfloat_batch(n, nb, int(g["seed"]))
    with Plan(n, FS, 1e9, max_batch=nb) as plan:
        assert np.array_equal(plan.window(), g["window_f32"])      # same window bits as the golden
        plan.submit_device(0, _to_dev(torch_cuda, x), nb)
        p, h, t = plan.collect(0)
    fig = tol.compare_spectra(p, g["db64"])
    assert fig["max_rel_power_vs_max_bin_mean"] < 5e-6
    assert len(h) == 0 and not t.any()


@pytest.mark.parametrize("n", [1024, 4096, 8192, 16384])
def test_hip_against_an_independent_float32_fft(torch_cuda, n):
    """The HIP path held directly to third-party arithmetic, without this repo's oracle in between: numpy's float32 multiply by
    the plan's window table, pocketfft (scipy.fft on complex64) and a float32 dB map (tests/test_oracle_vs_pocketfft.py) -- two
    independent single-precision FFTs, each a few 1e-6 from the float64 spectrum, within the 1e-5 bar of each other, and the
    same detections wherever the threshold is clear of every evaluated bin of both."""
    from tests.test_oracle_vs_pocketfft import float32_chain

    nb = 48
    x = synth.cfloat_batch(n, nb, seed=7 + n)
    with Plan(n, FS, 1e9, max_batch=nb) as plan:
        w = plan.window()
        plan.submit_device(0, _to_dev(torch_cuda, x), nb)
        p, _, _ = plan.collect(0)
    p_pf = float32_chain(x, w)
    print(n, "HIP vs pocketfft float32:", tol.compare_spectra(p, p_pf))
    keep = tol.evaluated_mask(n)
    thr = tol.pick_threshold(np.concatenate([p_pf, p]), n, start=10.0)   # guard band empty on BOTH spectra
    with Plan(n, FS, thr, max_batch=nb, max_hits=nb * n) as plan:
        plan.submit_device(0, _to_dev(torch_cuda, x), nb, 3e6 + 6e6 * np.arange(nb))
        _, h, _ = plan.collect(0, want_power=False)
    jj = (np.arange(n) + n // 2) % n
    want = [(b, int(i)) for b in range(nb) for i in np.nonzero(keep[jj] & (p_pf[b][jj] > np.float32(thr)))[0]]
    assert [(int(s), int(i)) for s, i in zip(h["seq_id"], h["i"])] == want and len(want) > 0


def test_one_slot_alternating_counts_only_and_records_collects(torch_cuda, oracle_mod):
    """The ordered list and its pinned copy exist ONCE per slot while regions and counts have two generations (scn_api.hip): a
    caller that collects counts only on one submit and reads records on the next -- through scn_collect, through scn_hits_view,
    with batches without any detection in between -- must always see the records of THAT submit (the ordering of the
    prefetch DMA against the slot's next compaction; the eager / on-demand decision across scn_collect and scn_hits_view)."""
    n, nb = 4096, 48
    batches = [synth.cfloat_batch(n, nb, seed=300 + k, sigma=0.05) for k in range(4)]
    quiet = (synth.cfloat_batch(n, nb, seed=299, sigma=0.05, max_tones=0) * np.float32(1e-3)).astype(np.complex64)   # nothing above the threshold
    fc = 1e9 + 6e6 * np.arange(nb)
    thr = tol.pick_threshold(np.concatenate([oracle_mod.Oracle(n, FS, 1e9).run(x)[0] for x in batches]), n, start=12.0)   # guard-band-free
    refs = [oracle_mod.Oracle(n, FS, thr).run(x, fc, np.arange(nb, dtype=np.uint64) + 1000 * k)[1] for k, x in enumerate(batches)]
    assert all(len(r) > 20 for r in refs)
    devs = [_to_dev(torch_cuda, x) for x in batches]
    dquiet = _to_dev(torch_cuda, quiet)
    with Plan(n, FS, thr, max_batch=nb, max_hits=1 << 16) as plan:
        def submit(k):
            plan.submit_device(0, devs[k], nb, fc, np.arange(nb, dtype=np.uint64) + 1000 * k)

        def same(h, k):
            # (spectra are held to the bar elsewhere; here: the records are those of submit k, in order)
            return len(h) == len(refs[k]) and np.array_equal(h["seq_id"], refs[k]["seq_id"]) and np.array_equal(h["i"], refs[k]["i"])

        order = [0, 1, 2, 3, 1, 0, 3, 2, 2, 1]
        for step, k in enumerate(order):       # ONE slot: every submit re-uses d_list / h_list of the one before
            submit(k)
            how = step % 3
            if how == 0:                         # counts only: nobody waits for this submit's list
                _, h, _ = plan.collect(0, want_power=False, want_hits=False)
                assert h is None and plan.last_n_hits == len(refs[k])
            elif how == 1:                       # records copied out by scn_collect
                _, h, _ = plan.collect(0, want_power=False, want_hits=True, hit_cap=1 << 16)
                assert same(h, k), (step, k)
            else:                                # counts from scn_collect, records read in place
                plan.collect(0, want_power=False, want_hits=False)
                assert same(plan.hits_view(0), k), (step, k)
        # a reader of the view stays an eager one across batches without detections (it never calls the view for those)
        for k in (0, 1):
            submit(k)
            plan.collect(0, want_power=False, want_hits=False)
            assert same(plan.hits_view(0), k)
            plan.submit_device(0, dquiet, nb, fc)
            plan.collect(0, want_power=False, want_hits=False)
            assert plan.last_n_hits == 0 and len(plan.hits_view(0)) == 0
        submit(2)
        plan.collect(0, want_power=False, want_hits=False)
        assert same(plan.hits_view(0), 2)
        # the view is a window on memory the next submit may overwrite: a copy taken now stays what it was
        keep = plan.hits_view(0).copy()
        submit(3)
        plan.collect(0, want_power=False, want_hits=False)
        assert same(plan.hits_view(0), 3) and same(keep, 2)


def test_pinned_double_buffered_submit(torch_cuda, oracle_mod):
    """scn_host_buffer + scn_submit on both slots (the replacement of sampleBuffer.cpp's
    staging), results identical to the device-resident path and to the oracle."""
    n, nb = 4096, 40
    xs = [synth.cfloat_batch(n, nb, seed=20 + s) for s in range(4)]
    thr = _clear_threshold(oracle_mod, n, xs, 9.0)
    o = oracle_mod.Oracle(n, FS, thr)
    with Plan(n, FS, thr, max_batch=nb, max_hits=1 << 16) as plan:
        views = [plan.host_buffer(s) for s in range(2)]
        assert views[0].nbytes == nb * n * 8 and views[0].ctypes.data != views[1].ctypes.data
        results = []
        # software pipeline: fill slot s while the other one is in flight
        for k, x in enumerate(xs):
            s = k & 1
            if k >= 2:
                results.append(plan.collect(s, hit_cap=1 << 16))
            views[s][:] = x.view(np.uint8).reshape(-1)
            fc = 100e6 + 6e6 * np.arange(nb) + k
            plan.submit(s, nb, fc, np.arange(k * nb, (k + 1) * nb, dtype=np.uint64))
        results.append(plan.collect(0, hit_cap=1 << 16))
        results.append(plan.collect(1, hit_cap=1 << 16))
        with pytest.raises(capi.ScannerError) as e:     # nothing pending any more
            plan.collect(0)
        assert e.value.status == capi.E_STATE
    for k, (p, h, t) in enumerate(results):
        fc = 100e6 + 6e6 * np.arange(nb) + k
        p_ref, h_ref, t_ref = o.run(xs[k], fc, np.arange(k * nb, (k + 1) * nb, dtype=np.uint64))
        tol.compare_spectra(p, p_ref)
        assert len(h_ref) > 0
        _assert_hits_equal(h, h_ref)      # the threshold's guard band is empty on every batch: bit-exact, always
        assert np.array_equal(t, t_ref)


@pytest.mark.parametrize("n,kind,enob,path", [(4096, capi.KIND_FLOAT_COMPLEX, 12, "device"),
                                              (8192, capi.KIND_SHORT_COMPLEX, 12, "device"),
                                              (4096, capi.KIND_SHORT_COMPLEX, 12, "device"),
                                              (4096, capi.KIND_FLOAT_COMPLEX, 12, "pinned")])
def test_overlapped_slots_give_identical_results(torch_cuda, n, kind, enob, path):
    """SCN_PLAN_OVERLAP_SLOTS puts the two slots on streams of their own so that consecutive launches
    overlap; the slots share nothing but read-only tables, so every spectrum bit and every hit must equal
    what the single-stream plan produces, in a pipeline that keeps both slots in flight."""
    torch = torch_cuda
    nb, rounds = 900, 6     # > one resident wave of workgroups (768 / 512): the integer kinds pull from the buffer queue
    raws = []
    for k in range(rounds):
        x = synth.cfloat_batch(n, nb, seed=300 + k)
        raws.append(synth.quantize(x, kind))
    fcs = [100e6 + 6e6 * np.arange(nb) + 7 * k for k in range(rounds)]
    seqs = [np.arange(k * nb, (k + 1) * nb, dtype=np.uint64) for k in range(rounds)]

    def run(flags):
        out = []
        with Plan(n, FS, 9.5, kind=kind, enob=enob, max_batch=nb, max_hits=1 << 18, flags=flags) as plan:
            if flags & capi.PLAN_OVERLAP_SLOTS:
                assert plan.slot_stream_handle(0) == plan.stream_handle != plan.slot_stream_handle(1)
            else:
                assert plan.slot_stream_handle(0) == plan.slot_stream_handle(1) == plan.stream_handle
            dev = [_to_dev(torch, r) for r in raws] if path == "device" else None
            views = [plan.host_buffer(s) for s in range(2)] if path == "pinned" else None
            for k in range(rounds):
                s = k & 1
                if k >= 2:
                    out.append(plan.collect(s, hit_cap=1 << 18))
                if path == "device":
                    plan.submit_device(s, dev[k], nb, fcs[k], seqs[k])
                else:
                    views[s][:] = np.ascontiguousarray(raws[k]).view(np.uint8).reshape(-1)
                    plan.submit(s, nb, fcs[k], seqs[k])
            out.append(plan.collect(rounds & 1, hit_cap=1 << 18))
            out.append(plan.collect((rounds + 1) & 1, hit_cap=1 << 18))
        return out

    base = capi.OUT_SPECTRUM | capi.OUT_HITS
    ref = run(base)
    got = run(base | capi.PLAN_OVERLAP_SLOTS)
    assert len(ref) == len(got) == rounds
    total = 0
    for (p0, h0, t0), (p1, h1, t1) in zip(ref, got):
        assert np.array_equal(p0.view(np.uint32), p1.view(np.uint32))       # bit-identical spectra
        assert h0.tobytes() == h1.tobytes() and np.array_equal(t0, t1)      # identical hit records, order, trigger flags
        total += len(h0)
    assert total > 100


@pytest.mark.parametrize("n,kind,enob,flags", [
    (4096, capi.KIND_FLOAT_COMPLEX, 12, 0), (4096, capi.KIND_SHORT_COMPLEX, 12, capi.PLAN_OVERLAP_SLOTS),
    (8192, capi.KIND_BYTE_COMPLEX, 8, 0), (16384, capi.KIND_SHORT_COMPLEX, 12, 0), (512, capi.KIND_SHORT, 12, 0),
])
def test_all_slots_in_flight(torch_cuda, n, kind, enob, flags):
    """SCN_NUM_SLOTS submits in flight (the records pipeline of bench.py: kernel, list kernels, DMA and the caller's copy of
    up to four submits overlap): every spectrum bit, every ordered record and every trigger flag must equal what the same
    batches give one at a time on slot 0 -- in a pipeline that collects a slot only right before it is reused, through the
    copying collect and through scn_hits_view."""
    torch = torch_cuda
    assert capi.NUM_SLOTS >= 4
    nb, rounds = 700, 11    # (the integer kinds pull from the buffer queue, whose bases the host tracks per slot)
    raws = [synth.quantize(synth.cfloat_batch(n, nb, seed=1300 + k), kind) for k in range(rounds)]
    sizes = [nb, nb - 1, 3, nb, 650, nb, 1, nb, 333, nb, nb]
    fcs = [100e6 + 6e6 * np.arange(nb) + 11 * k for k in range(rounds)]
    seqs = [np.arange(k * nb, (k + 1) * nb, dtype=np.uint64) for k in range(rounds)]
    fl = capi.OUT_SPECTRUM | capi.OUT_HITS | flags

    def take(plan, s, view):
        if not view:
            return plan.collect(s, hit_cap=1 << 18)
        p, _, t = plan.collect(s, want_hits=False)
        return p, plan.hits_view(s).copy(), t

    with Plan(n, FS, 9.5, kind=kind, enob=enob, max_batch=nb, max_hits=1 << 18, flags=fl) as plan:
        dev = [_to_dev(torch, r) for r in raws]
        ref = []
        for k in range(rounds):
            plan.submit_device(0, dev[k], sizes[k], fcs[k][:sizes[k]], seqs[k][:sizes[k]])
            ref.append(plan.collect(0, hit_cap=1 << 18))
        for view in (False, True):
            got = []
            for k in range(rounds):
                s = k % capi.NUM_SLOTS
                if k >= capi.NUM_SLOTS:
                    got.append(take(plan, s, view))
                plan.submit_device(s, dev[k], sizes[k], fcs[k][:sizes[k]], seqs[k][:sizes[k]])
            for j in range(capi.NUM_SLOTS):  # oldest first
                got.append(take(plan, (rounds + j) % capi.NUM_SLOTS, view))
            assert len(got) == rounds
            total = 0
            for (p0, h0, t0), (p1, h1, t1) in zip(ref, got):
                assert np.array_equal(p0.view(np.uint32), p1.view(np.uint32))
                assert h0.tobytes() == h1.tobytes() and np.array_equal(t0, t1)
                total += len(h0)
            assert total > 100


@pytest.mark.parametrize("n,kind,enob,dc", [(4096, capi.KIND_SHORT_COMPLEX, 12, False), (4096, capi.KIND_BYTE_COMPLEX, 8, False),
                                            (1024, capi.KIND_SHORT, 12, True), (8192, capi.KIND_BYTE_COMPLEX, 8, False)])
def test_buffer_queue_over_many_launches(torch_cuda, oracle_mod, n, kind, enob, dc):
    """The integer formats' workgroups take their buffers from a device-side queue whose heads are never reset
    (the host tracks their base per slot).  Batches larger than one resident wave of workgroups, of changing
    size, on both slots, many launches in a row: every buffer of every launch must be processed exactly once --
    spectra and hit lists against the oracle."""
    torch = torch_cuda
    sizes = [1800, 2047, 769, 5, 1500, 3000, 1, 2500]
    rng_seed = 900
    with Plan(n, FS, 9.5, kind=kind, enob=enob, correct_dc=dc, max_batch=max(sizes), max_hits=1 << 19) as plan:
        o = oracle_mod.Oracle(n, FS, 9.5, kind=kind, enob=enob, correct_dc=dc)
        pending = {}
        for k, nb in enumerate(sizes):
            s = k & 1
            if s in pending:
                _check_launch(plan, o, n, s, *pending.pop(s))
            raw = synth.quantize(synth.cfloat_batch(n, nb, seed=rng_seed + k), kind)
            if dc:  # a positive offset: the integer mean is then an ordinary small number (the negative-sum quirk of
                raw = (raw + 37).astype(raw.dtype)   # utility.cpp:77-78 has its own test, test_dc_quirk_negative_mean)
            fc = 100e6 + 6e6 * np.arange(nb) + k
            seq = np.arange(1000 * k, 1000 * k + nb, dtype=np.uint64)
            plan.submit_device(s, _to_dev(torch, raw), nb, fc, seq)
            pending[s] = (raw, fc, seq)
        for s in sorted(pending):
            _check_launch(plan, o, n, s, *pending[s])


def _check_launch(plan, o, n, slot, raw, fc, seq):
    p, h, t = plan.collect(slot, hit_cap=1 << 19)
    p_ref, h_ref, t_ref = o.run(raw, fc, seq, threads=8)
    tol.compare_spectra(p, p_ref)
    # thousands of buffers per launch: no threshold has an empty guard band on all of them, so the records outside the
    # band are demanded bit for bit (always), and the band's population bounds what may differ
    if _assert_hits_equal_outside_guard(h, h_ref, p_ref, n, 9.5) == 0:
        _assert_hits_equal(h, h_ref)
        assert np.array_equal(t, t_ref)


# ---------------------------------------------------------------------------------------
# the other FFT sizes: BASELINE C1 (1024-pt, one cfloat buffer), C3 (8192-pt int16), 2048
# ---------------------------------------------------------------------------------------
def test_c1_known_answer_tone_1024(torch_cuda, oracle_mod):
    """BASELINE config C1 + SURVEY 8c known answer: one 1024-pt cfloat buffer, five hits around 101 MHz."""
    g = np.load(os.path.join(GOLD, "known_answer_tone.npz"))
    n = int(g["n"])
    x = (0.5 * np.exp(2j * np.pi * 128 * np.arange(n) / n)).astype(np.complex64)[None]
    with Plan(n, int(g["sample_rate"]), float(g["threshold"]), max_batch=1) as plan:
        plan.submit_device(0, _to_dev(torch_cuda, x), 1, [float(g["center_freq"])], [42])
        p, h, t = plan.collect(0)
    assert h["i"].tolist() == g["hit_i"].tolist() and h["freq_hz"].tolist() == g["hit_freq"].tolist()
    assert np.all(h["seq_id"] == 42) and t.tolist() == [0]
    assert np.abs(h["power_db"] - g["hit_db64"]).max() < 1e-5
    assert "freq %d power_db %f" % (h["freq_hz"][2], h["power_db"][2]) == "freq 100999680 power_db 22.636375"


@pytest.mark.parametrize("n", [1024, 2048, 8192, 16384])
def test_sizes_cfloat_vs_oracle_and_golden(torch_cuda, oracle_mod, n):
    nb = {1024: 300, 2048: 130, 8192: 70, 16384: 300}[n]          # not a multiple of the resident grid
    x = synth.cfloat_batch(n, nb, seed=40 + n)
    fc = 3e6 + 6e6 * np.arange(nb)
    p_ref, _, _ = oracle_mod.Oracle(n, FS, 1e9).run(x, threads=4)
    thr = tol.pick_threshold(p_ref, n, start=8.0)
    (p, h, t), (p_ref, h_ref, t_ref) = _run_both(torch_cuda, oracle_mod, n, capi.KIND_FLOAT_COMPLEX, x, fc, None, thr)
    print(n, tol.compare_spectra(p, p_ref))
    assert len(h_ref) > 20
    _assert_hits_equal(h, h_ref)
    assert np.array_equal(t, t_ref)
    gp = os.path.join(GOLD, f"spectrum_n{n}.npz")
    if os.path.exists(gp):
        g = np.load(gp)
        xg = synth.cfloat_batch(n, int(g["n_buffers"]), int(g["seed"]))
        with Plan(n, FS, 1e9, max_batch=8) as plan:
            plan.submit_device(0, _to_dev(torch_cuda, xg), len(xg))
            pg, _, _ = plan.collect(0)
        assert tol.compare_spectra(pg, g["db64"])["max_rel_power_vs_max_bin_mean"] < 5e-6


@pytest.mark.parametrize("n,kind,enob,dc", [
    (8192, capi.KIND_SHORT_COMPLEX, 12, False),    # BASELINE config C3
    (8192, capi.KIND_SHORT_COMPLEX, 12, True),
    (8192, capi.KIND_BYTE_COMPLEX, 8, False),
    (8192, capi.KIND_SHORT, 12, True),
    (1024, capi.KIND_SHORT_COMPLEX, 12, True),
    (1024, capi.KIND_BYTE_COMPLEX, 8, True),
    (2048, capi.KIND_SHORT, 12, False),
    (2048, capi.KIND_SHORT_COMPLEX, 16, True),
    (16384, capi.KIND_SHORT_COMPLEX, 12, False),   # the largest size that fits the LDS (scn_fft16k_kernel)
    (16384, capi.KIND_SHORT_COMPLEX, 12, True),
    (16384, capi.KIND_BYTE_COMPLEX, 8, True),
    (16384, capi.KIND_SHORT, 14, False),
])
def test_sizes_integer_kinds(torch_cuda, oracle_mod, n, kind, enob, dc):
    nb = 37
    x = synth.cfloat_batch(n, nb, seed=50 + n, sigma=0.1) + np.complex64(0.015 - 0.01j)
    raw = synth.quantize(x, kind)
    fc = 9e8 + 6e6 * np.arange(nb)
    p_ref, _, _ = oracle_mod.Oracle(n, FS, 1e9, kind=kind, enob=enob, correct_dc=dc).run(raw, threads=4)
    thr = tol.pick_threshold(p_ref, n, start=float(np.quantile(p_ref[:, tol.evaluated_mask(n)], 0.999)))
    (p, h, t), (p_ref, h_ref, t_ref) = _run_both(torch_cuda, oracle_mod, n, kind, raw, fc, None, thr, enob, dc)
    tol.compare_spectra(p, p_ref)
    assert len(h_ref) > 0
    _assert_hits_equal(h, h_ref)
    assert np.array_equal(t, t_ref)


@pytest.mark.parametrize("n,kind,enob,dc", [
    (16, capi.KIND_FLOAT_COMPLEX, 12, False), (64, capi.KIND_SHORT_COMPLEX, 12, True), (256, capi.KIND_FLOAT_COMPLEX, 12, False),
    (512, capi.KIND_BYTE_COMPLEX, 8, True), (512, capi.KIND_SHORT, 12, False), (32768, capi.KIND_FLOAT_COMPLEX, 12, False),
    (32768, capi.KIND_SHORT_COMPLEX, 12, True), (65536, capi.KIND_FLOAT_COMPLEX, 12, False), (65536, capi.KIND_BYTE_COMPLEX, 8, False),
])
def test_generic_sizes_vs_oracle(torch_cuda, oracle_mod, n, kind, enob, dc):
    """The reference plans any --count (fft.cpp:4-11).  Powers of two outside 1024 ... 16384: 16 ... 128 and 32768 run through the
    staged path of scn_generic.hip, 256 / 512 through the several-buffers-per-workgroup kernel, 65536 through the four-step pair
    of scn_big.hip: same spectra (to the bar), same hit lists, same trigger flags."""
    nb = {16: 200, 64: 150, 256: 90, 512: 75, 32768: 9, 65536: 5}[n]
    x = synth.cfloat_batch(n, nb, seed=70 + n % 1000, sigma=0.1)
    raw = synth.quantize(x, kind) if kind != capi.KIND_FLOAT_COMPLEX else x
    if dc:  # a DC offset to remove -- clipped, not wrapped: a wrapped int8 buffer is dominated by its jumps, the threshold below
        info = np.iinfo(raw.dtype)  # would then sit far under the buffer mean, where two float32 FFTs may disagree on a bin
        raw = np.clip(raw.astype(np.int32) + 9, info.min, info.max).astype(raw.dtype)
    fc = 2.4e9 + 6e6 * np.arange(nb)
    p_ref, _, _ = oracle_mod.Oracle(n, FS, 1e9, kind=kind, enob=enob, correct_dc=dc).run(raw, threads=4)
    ev = tol.evaluated_mask(n)
    thr = tol.pick_threshold(p_ref, n, start=float(np.quantile(p_ref[:, ev], 0.97))) if ev.any() else 0.0
    (p, h, t), (p_ref, h_ref, t_ref) = _run_both(torch_cuda, oracle_mod, n, kind, raw, fc, None, thr, enob, dc, max_hits=nb * n)
    print(n, tol.compare_spectra(p, p_ref))
    assert len(h_ref) > 0 or not ev.any()
    _assert_hits_equal(h, h_ref)
    assert np.array_equal(t, t_ref)


@pytest.mark.parametrize("n,kind,enob,dc", [
    (256, capi.KIND_SHORT_COMPLEX, 12, True), (256, capi.KIND_SHORT, 12, False), (256, capi.KIND_BYTE_COMPLEX, 8, False),
    (512, capi.KIND_FLOAT_COMPLEX, 12, False), (512, capi.KIND_SHORT_COMPLEX, 12, True), (512, capi.KIND_SHORT, 12, True),
    (65536, capi.KIND_SHORT_COMPLEX, 12, False), (65536, capi.KIND_SHORT, 12, False),
    # DC removal on the four-step path: scn_big_dc_kernel takes the buffer's integer sums first
    (65536, capi.KIND_SHORT_COMPLEX, 12, True), (65536, capi.KIND_BYTE_COMPLEX, 8, True), (32768, capi.KIND_SHORT, 12, True),
    (32768, capi.KIND_FLOAT_COMPLEX, 12, False), (32768, capi.KIND_SHORT, 12, False), (32768, capi.KIND_BYTE_COMPLEX, 8, False),
])
@pytest.mark.parametrize("flags", ["both", "hits", "spectrum"])
def test_round3_kernels_formats_and_output_modes(torch_cuda, oracle_mod, n, kind, enob, dc, flags):
    """The kernels new in round 3 -- several buffers per workgroup at 256 / 512 points (scn_fft_small_kernel), the four-step
    pairs for plain 65536- and 32768-point plans (scn_big.hip) -- for every wire format, with batch sizes that leave the last workgroup's
    buffer slots partly empty, in all three output modes: spectra to the bar, hit lists bit for bit."""
    nb = {256: 16 * 9 + 5, 512: 8 * 11 + 3, 65536: 4, 32768: 7}[n]
    x = synth.cfloat_batch(n, nb, seed=170 + n % 1000, sigma=0.1)
    raw = synth.quantize(x, kind) if kind != capi.KIND_FLOAT_COMPLEX else x
    if dc:
        info = np.iinfo(raw.dtype)
        raw = np.clip(raw.astype(np.int32) + 9, info.min, info.max).astype(raw.dtype)
    fc = 433e6 + 6e6 * np.arange(nb)
    p_ref, _, _ = oracle_mod.Oracle(n, FS, 1e9, kind=kind, enob=enob, correct_dc=dc).run(raw, threads=4)
    ev = tol.evaluated_mask(n)
    thr = tol.pick_threshold(p_ref, n, start=float(np.quantile(p_ref[:, ev], 0.97)))
    p_ref, h_ref, t_ref = oracle_mod.Oracle(n, FS, thr, kind=kind, enob=enob, correct_dc=dc).run(raw, fc, threads=4)
    assert len(h_ref) > 0
    fl = {"both": capi.OUT_SPECTRUM | capi.OUT_HITS, "hits": capi.OUT_HITS, "spectrum": capi.OUT_SPECTRUM}[flags]
    with Plan(n, FS, thr, kind=kind, enob=enob, correct_dc=dc, max_batch=nb, max_hits=nb * n, flags=fl) as plan:
        plan.submit_device(1, _to_dev(torch_cuda, raw), nb, fc)
        p, h, t = plan.collect(1, hit_cap=nb * n)
    if flags != "hits":
        tol.compare_spectra(p, p_ref)
    else:
        assert p is None
    if flags != "spectrum":
        assert len(h) == len(h_ref)
        for f in ("seq_id", "i", "freq_hz"):
            assert np.array_equal(h[f], h_ref[f]), f
        assert np.array_equal(t, t_ref)
        if flags == "both":   # a record carries the float the spectrum holds (spectra themselves are held to the bar above)
            jj = (h["i"].astype(np.int64) + n // 2) % n
            assert np.array_equal(h["power_db"], p[h["seq_id"].astype(np.int64), jj])
        else:                 # no spectrum to hold them to: the bar's dB form on the bins it covers, a loose bound elsewhere
            big = tol.db_to_power(h_ref["power_db"]) >= tol.db_to_power(p_ref).mean(axis=-1)[h_ref["seq_id"].astype(np.int64)]
            err = np.abs(h["power_db"].astype(np.float64) - h_ref["power_db"])
            assert np.all(err[big] <= tol.DB_REL * np.abs(h_ref["power_db"][big]) + tol.DB_ABS) and err.max(initial=0) < 0.2


def test_strong_tone_accuracy_16384(torch_cuda, oracle_mod):
    """The accuracy tail of the largest fused size (1.05e-5 on one of 3072 strong-tone buffers, ON the bar).
    1024 buffers per wire format with up to four tones of amplitude 0.05 .. 0.5 in sigma = 0.05 noise (peak / mean power up to
    7e3): with pass 3 in double and the exact half of the dB map the metric's maximum stays well inside 1e-5 -- asserted at
    8e-6, measured 2.9e-6 .. 4.0e-6 (scripts/acc16k.py; the float build reads 4.9e-6 .. 7.5e-6, pocketfft's float32 transform
    on the same buffers 7.0e-6)."""
    n, nb = 16384, 256
    for kind, enob in ((capi.KIND_BYTE_COMPLEX, 8), (capi.KIND_SHORT_COMPLEX, 12), (capi.KIND_FLOAT_COMPLEX, 12)):
        worst = 0.0
        for seed in range(4):
            x = synth.cfloat_batch(n, nb, seed=900 + seed)
            raw = synth.quantize(x, kind)
            p_ref, _, _ = oracle_mod.Oracle(n, FS, 1e9, kind=kind, enob=enob).run(raw, threads=8)
            with Plan(n, FS, 1e9, kind=kind, enob=enob, max_batch=nb) as plan:
                plan.submit_device(0, _to_dev(torch_cuda, raw), nb)
                p, _, _ = plan.collect(0)
            worst = max(worst, tol.compare_spectra(p, p_ref)["max_rel_power_vs_max_bin_mean"])
        print("16384-point strong-tone buffers, kind", kind, "max", worst)
        assert worst <= 8e-6, (kind, worst)


@pytest.mark.parametrize("n,kind,enob,dc", [
    (30, capi.KIND_FLOAT_COMPLEX, 12, False), (1000, capi.KIND_SHORT_COMPLEX, 12, True), (1023, capi.KIND_FLOAT_COMPLEX, 12, False),
    (3000, capi.KIND_BYTE_COMPLEX, 8, False), (4097, capi.KIND_SHORT, 12, True), (6000, capi.KIND_FLOAT_COMPLEX, 12, False),
    (17, capi.KIND_SHORT_COMPLEX, 12, False), (20000, capi.KIND_BYTE_COMPLEX, 8, False), (32767, capi.KIND_FLOAT_COMPLEX, 12, False),
    (65535, capi.KIND_SHORT_COMPLEX, 12, False),   # the ends of the range (65535: a 131072-point transform)
])
def test_non_power_of_two_sizes_vs_oracle(torch_cuda, oracle_mod, n, kind, enob, dc):
    """--count is any integer in the reference (FFTW plans it, fft.cpp:4-11).  Sizes that are not powers of two run
    Bluestein's algorithm over the staged path; the oracle evaluates the DFT sum itself in double for them.  Odd sizes
    exercise the general form of the mask: j = (i + N/2) % N with an integer N/2 (process.cpp:45-47)."""
    nb = 24 if n < 8000 else 3 if n < 40000 else 2      # (the oracle's direct DFT is O(N^2) per buffer)
    x = synth.cfloat_batch(n, nb, seed=80 + n % 1000, sigma=0.1)
    raw = synth.quantize(x, kind) if kind != capi.KIND_FLOAT_COMPLEX else x
    if dc:
        info = np.iinfo(raw.dtype)
        raw = np.clip(raw.astype(np.int32) + 21, info.min, info.max).astype(raw.dtype)   # (a positive mean: the negative-sum quirk
        # turns a buffer into one huge constant plus cancellation residue -- test_dc_quirk_negative_mean covers it where it belongs)
    fc = 915e6 + 6e6 * np.arange(nb)
    p_ref, _, _ = oracle_mod.Oracle(n, FS, 1e9, kind=kind, enob=enob, correct_dc=dc).run(raw, threads=4)
    ev = tol.evaluated_mask(n)
    thr = tol.pick_threshold(p_ref, n, start=float(np.quantile(p_ref[:, ev][np.isfinite(p_ref[:, ev])], 0.97)))
    (p, h, t), (p_ref, h_ref, t_ref) = _run_both(torch_cuda, oracle_mod, n, kind, raw, fc, None, thr, enob, dc, max_hits=nb * n)
    print(n, tol.compare_spectra(p, p_ref))
    assert len(h_ref) > 0
    _assert_hits_equal(h, h_ref)
    assert np.array_equal(t, t_ref)


def test_generic_size_double_buffered_and_windows(torch_cuda, oracle_mod):
    """Both slots in flight at a generic size, every evaluated bin a hit (the region / compaction capacity at 65536 points:
    49 145 evaluated bins per buffer, a 2048-word bitmap per wave), the list walked in windows."""
    n, nb = 65536, 3
    xs = [synth.cfloat_batch(n, nb, seed=90 + s, sigma=0.1) for s in range(2)]
    fc = np.array([1e9, 2e9, 3e9])
    with Plan(n, FS, -200.0, max_batch=nb, max_hits=4096, flags=capi.OUT_HITS) as plan:
        for s in range(2):
            plan.submit_device(s, _to_dev(torch_cuda, xs[s]), nb, fc)
        for s in range(2):
            _, h_ref, t_ref = oracle_mod.Oracle(n, FS, -200.0).run(xs[s], fc, threads=4)
            _, h, t = plan.collect(s, want_power=False)
            assert len(h_ref) == nb * tol.evaluated_mask(n).sum() == len(h)
            _assert_hits_equal(h, h_ref)
            assert np.array_equal(t, t_ref) and t.all()


@pytest.mark.parametrize("n", [1024, 8192, 16384])
def test_sizes_all_bins_hit_mask(torch_cuda, oracle_mod, n):
    """mask edges (DC window, use-band) and the i <-> j mapping at the other sizes"""
    nb = 3
    x = synth.cfloat_batch(n, nb, seed=60 + n, sigma=0.1)
    fc = np.array([1e9, 2e9, 3e9])
    (p, h, t), (p_ref, h_ref, t_ref) = _run_both(torch_cuda, oracle_mod, n, capi.KIND_FLOAT_COMPLEX, x, fc, None,
                                                 -200.0, max_hits=nb * n * 2)
    assert len(h) == nb * tol.evaluated_mask(n).sum()
    _assert_hits_equal(h, h_ref)
    assert np.array_equal(t, t_ref)        # 1024: 762 hits <= 1047 -> 0 ; 8192: 6138 -> 1


def test_c3_full_batch_8192_int16(torch_cuda, oracle_mod):
    """BASELINE config C3 at a full batch: 4096 x 8192-pt int16 buffers, every buffer vs the oracle."""
    n, nb = 8192, 4096
    xd = synth.cfloat_batch_torch(n, nb, seed=3, device="cuda", sigma=0.1)
    raw_d = torch_cuda.clamp(torch_cuda.round(xd * 2047.0), -2048, 2047).to(torch_cuda.int16).contiguous()
    del xd
    raw = raw_d.cpu().numpy()
    fc = 3e6 + 6e6 * np.arange(nb)
    o = oracle_mod.Oracle(n, FS, 12.0, kind=capi.KIND_SHORT_COMPLEX, enob=12)
    p_ref, h_ref, t_ref = o.run(raw, fc, threads=8)
    with Plan(n, FS, 12.0, kind=capi.KIND_SHORT_COMPLEX, enob=12, max_batch=nb, max_hits=1 << 22) as plan:
        plan.submit_device(0, raw_d, nb, fc)
        p, h, t = plan.collect(0, hit_cap=1 << 22)
    print("C3 full batch vs oracle:", tol.compare_spectra(p, p_ref))
    m = tol.evaluated_mask(n)
    near = np.zeros((nb, n), bool)
    near[:, m] = np.abs(p_ref[:, m] - 12.0) < tol.GUARD_DB
    key = lambda a: a["seq_id"].astype(np.int64) * n + a["i"]       # noqa: E731
    jj = lambda a: (a["i"].astype(np.int64) + n // 2) % n           # noqa: E731
    assert np.array_equal(key(h[~near[h["seq_id"].astype(np.int64), jj(h)]]),
                          key(h_ref[~near[h_ref["seq_id"].astype(np.int64), jj(h_ref)]]))
    assert np.array_equal(t, t_ref)


# ---------------------------------------------------------------------------------------
# integer wire formats (K1a-c) incl. the reference's quirks
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("kind,enob,dc", [
    (capi.KIND_SHORT_COMPLEX, 12, False),
    (capi.KIND_SHORT_COMPLEX, 12, True),
    (capi.KIND_SHORT_COMPLEX, 16, False),     # int16_t max wraps to -32768: sign-flipped samples
    (capi.KIND_SHORT, 12, False),
    (capi.KIND_SHORT, 12, True),
    (capi.KIND_BYTE_COMPLEX, 8, False),       # int8_t max wraps to -128
    (capi.KIND_BYTE_COMPLEX, 8, True),
    (capi.KIND_BYTE_COMPLEX, 7, False),
])
def test_integer_kinds_4096(torch_cuda, oracle_mod, kind, enob, dc):
    n, nb = 4096, 24
    x = synth.cfloat_batch(n, nb, seed=3, sigma=0.1)
    x += np.complex64(0.02 + 0.01j)                       # positive DC offset for the DC path
    raw = synth.quantize(x, kind)
    fc = 2.4e9 + 6e6 * np.arange(nb)
    o = oracle_mod.Oracle(n, FS, 1e9, kind=kind, enob=enob, correct_dc=dc)
    p_ref, _, _ = o.run(raw, threads=4)
    thr = tol.pick_threshold(p_ref, n, start=float(np.quantile(p_ref, 0.999)))
    (p, h, t), (p_ref, h_ref, t_ref) = _run_both(torch_cuda, oracle_mod, n, kind, raw, fc, None, thr, enob, dc)
    tol.compare_spectra(p, p_ref)
    assert len(h_ref) > 0
    _assert_hits_equal(h, h_ref)
    assert np.array_equal(t, t_ref)


def _negative_sum_buffers(n, nb, kind, seed):
    """raw integer buffers whose I and Q sums are NEGATIVE on both rails (the `int32 /= uint32` of utility.cpp:77-78 then yields a
    'mean' of ~2^32 / n), mixed with buffers that are negative on one rail only and with ordinary positive-mean ones, so that the
    lanes of one wave / the slots of one workgroup hold different cases side by side"""
    rng = np.random.default_rng(seed)
    lo, hi = (-100, 40) if kind == capi.KIND_BYTE_COMPLEX else (-300, 100)
    raw = rng.integers(lo, hi, size=(nb, n, 2)).astype(np.int8 if kind == capi.KIND_BYTE_COMPLEX else np.int16)
    raw[2::5] = -raw[2::5]                    # positive on both rails
    raw[3::5, :, 1] = -raw[3::5, :, 1]        # I negative, Q positive
    raw[4::5, :, 0] = -raw[4::5, :, 0]        # I positive, Q negative
    if kind == capi.KIND_SHORT:
        raw = np.ascontiguousarray(np.moveaxis(raw, -1, -2))   # planar: I[n] then Q[n]
    return raw


# one size per kernel family, each with its own DC reduction: tiny (a segmented shuffle over R lanes), small (over T lanes),
# narrow (wave sums through LDS), 8192 (two samples per lane and load), 16384 (half a buffer prefetched), the four-step pair
# (scn_big_dc_kernel: a pass of its own over the raw samples), Bluestein (scn_gen_load_kernel)
@pytest.mark.parametrize("kind,enob", [(capi.KIND_SHORT_COMPLEX, 12), (capi.KIND_SHORT, 12), (capi.KIND_BYTE_COMPLEX, 8)],
                         ids=["int16", "int16planar", "int8"])
@pytest.mark.parametrize("n", [64, 256, 1024, 4096, 8192, 16384, 32768, 1000])
def test_dc_quirk_negative_mean(torch_cuda, oracle_mod, n, kind, enob):
    """utility.cpp:77-78: int32 /= uint32 turns a negative sum into a huge positive 'mean'.  Every kernel family must reproduce the
    same (nonsensical) samples, hence the same spectrum and the same detections -- bit for bit wherever the spectrum tolerance
    itself cannot move a bin across the threshold (tolerances.flip_unsafe)."""
    nb = {64: 331, 256: 83, 1024: 41, 4096: 13, 8192: 11, 16384: 7, 32768: 5, 1000: 12}[n]
    raw = _negative_sum_buffers(n, nb, kind, seed=4 + n)
    o = oracle_mod.Oracle(n, FS, 1e9, kind=kind, enob=enob, correct_dc=True)
    c = o.convert(raw[0].reshape(-1))
    assert abs(c.real).min() > 30 and abs(c.imag).min() > 30         # the quirk is in play: a 'mean' of ~2^32 / n on both rails
    assert abs(o.convert(raw[2].reshape(-1)).real).max() < 2          # ... and an ordinary buffer beside it
    p_ref, _, _ = o.run(raw, threads=4)
    ev = tol.evaluated_mask(n)
    # (a quirk buffer is one huge constant: by the parity metric -- relative to max(bin, buffer mean) -- ALL its other bins are
    #  within the tolerance of any threshold, so bit-exact records can only be demanded of the ordinary buffers beside them; the
    #  threshold makes half of THEIR evaluated bins detections)
    thr = float(np.median(p_ref[2::5][:, ev]))
    fc = 88e6 + 6e6 * np.arange(nb)
    (p, h, t), (p_ref, h_ref, t_ref) = _run_both(torch_cuda, oracle_mod, n, kind, raw, fc, None, thr, enob, True, max_hits=nb * n)
    print(n, tol.compare_spectra(p, p_ref))
    unsafe = tol.flip_unsafe(p_ref, thr)

    def safe(a):
        return a[~unsafe[a["seq_id"].astype(np.int64), (a["i"].astype(np.int64) + n // 2) % n]]

    g, r = safe(h), safe(h_ref)
    assert len(r) > len(p_ref[2::5]) * ev.sum() // 4 and len(h_ref) > len(r) and len(g) == len(r), (len(g), len(r), len(h), len(h_ref))
    for f in ("seq_id", "i", "freq_hz"):
        assert np.array_equal(g[f], r[f]), f
    cnt = np.bincount(h_ref["seq_id"].astype(np.int64), minlength=nb)
    trig_safe = (cnt + unsafe[:, ev].sum(axis=1) <= 1047) | (cnt - unsafe[:, ev].sum(axis=1) > 1047)
    assert np.array_equal(t[trig_safe], t_ref[trig_safe])


@pytest.mark.parametrize("kind,enob", [(capi.KIND_SHORT_COMPLEX, 12), (capi.KIND_SHORT, 12), (capi.KIND_BYTE_COMPLEX, 8)],
                         ids=["int16", "int16planar", "int8"])
@pytest.mark.parametrize("n", [8192, 1004])   # the streaming one-wave-per-buffer form, and the per-sample form (n not a multiple of 8)
def test_dc_quirk_negative_mean_time_domain(torch_cuda, oracle_mod, n, kind, enob):
    """the same quirk through the time-domain kernels' own DC reduction (process.cpp:203-237 behind utility.cpp:70-79)"""
    nb = 40
    raw = _negative_sum_buffers(n, nb, kind, seed=9 + n)
    o = oracle_mod.Oracle(n, FS, 0.0, kind=kind, enob=enob, correct_dc=True)
    flat = raw.reshape(nb, -1)
    ref = [o.time_domain(o.convert(flat[b]), threshold=20.0) for b in range(nb)]
    ref_max, ref_min = np.array([r[1] for r in ref], np.float32), np.array([r[2] for r in ref], np.float32)
    ref_hit = np.array([r[0] for r in ref], np.uint8)
    assert ref_max[0] > 22 and ref_max[2] < 10 and 0 < ref_hit.sum() < nb    # quirk buffers are ~2^32 / n in size, the others are not
    with Plan(n, FS, 20.0, kind=kind, enob=enob, correct_dc=True, max_batch=64, mode=capi.MODE_TIME_DOMAIN) as plan:
        plan.submit_device(0, _to_dev(torch_cuda, raw), nb, np.arange(nb) * 1e6)
        mx, mn, ab = plan.collect_time_domain(0)
    fin = np.isfinite(ref_min)
    assert np.array_equal(np.isfinite(mn), fin)
    assert np.abs(mx - ref_max).max() < 1e-4 and np.abs(mn[fin] - ref_min[fin]).max(initial=0) < 2e-3
    assert np.array_equal(ab, ref_hit)


# ---------------------------------------------------------------------------------------
# K5 semantics: mask, strictness, ordering, trigger, capacity
# ---------------------------------------------------------------------------------------
def test_every_evaluated_bin_hits_and_trigger(torch_cuda, oracle_mod):
    n, nb = 4096, 5
    x = synth.cfloat_batch(n, nb, seed=7, sigma=0.1)
    fc = np.array([0.0, 4e6, 433.92e6, 5.9e9, 100e6])       # fc=0 -> start frequency negative is avoided below
    fc[0] = 4e6
    (p, h, t), (p_ref, h_ref, t_ref) = _run_both(torch_cuda, oracle_mod, n, capi.KIND_FLOAT_COMPLEX, x, fc,
                                                 np.array([5, 4, 3, 2, 1], np.uint64), -200.0, max_hits=nb * n)
    m = tol.evaluated_mask(n)
    assert len(h) == nb * m.sum() == nb * 3066
    _assert_hits_equal(h, h_ref)                         # includes order: buffer-major, then i
    assert t.tolist() == [1] * nb == t_ref.tolist()       # 3066 > 1047 (process.cpp:62)
    # the values reported ARE the spectrum values at j = (i + N/2) % N
    j = (h["i"].astype(np.int64) + n // 2) % n
    b = np.repeat(np.arange(nb), m.sum())
    assert np.array_equal(h["power_db"], p[b, j])


@pytest.mark.parametrize("use_bw,dc_bins", [(0.5, 8), (1.0, 0), (0.9, 1), (0.75, 4)])
@pytest.mark.parametrize("n", [64, 256, 1024, 4096, 8192, 16384, 3000, 12000, 32768, 1023])
def test_mask_parameters_across_kernel_families(torch_cuda, oracle_mod, n, use_bw, dc_bins):
    """process.cpp:46-52 with other parameters than the CLI's defaults (useBandWidth 0.75, dcIgnoreWindow 4): every kernel family
    builds its keep mask from the plan's dc_ignore_bins / i_lo / i_hi -- tiny, small, narrow, 8192, 16384, both mixed-radix forms,
    the four-step pair, Bluestein.  With a threshold under every bin the hit list IS the mask: (seq_id, i, freq_hz) bit for bit,
    and exactly mask.sum() records per buffer (use_bandwidth 1.0: U = N/2, every i kept; dc_ignore 0: no DC window)."""
    nb = 4 if n >= 8192 else 9
    x = synth.cfloat_batch(n, nb, seed=11 + n, sigma=0.1)
    fc = 1.2e9 + 6e6 * np.arange(nb)
    o = oracle_mod.Oracle(n, FS, -200.0, use_bandwidth=use_bw, dc_ignore_bins=dc_bins)
    _, h_ref, t_ref = o.run(x, fc, None, threads=4)
    m = tol.evaluated_mask(n, use_bw, dc_bins)
    assert len(h_ref) == nb * int(m.sum()) > 0
    with Plan(n, FS, -200.0, max_batch=nb, max_hits=nb * n, use_bandwidth=use_bw, dc_ignore_bins=dc_bins, flags=capi.OUT_HITS) as plan:
        plan.submit_device(0, _to_dev(torch_cuda, x), nb, fc)
        _, h, t = plan.collect(0, hit_cap=nb * n)
    _assert_hits_equal(h, h_ref)
    assert np.array_equal(t, t_ref)


def test_trigger_threshold_edge(torch_cuda, oracle_mod):
    """trigger = hits > 1047, strictly (process.cpp:62): build spectra with exactly 1047 / 1048 hits."""
    n = 4096
    k = np.arange(n)
    x = np.zeros((2, n), np.complex64)
    rng = np.random.default_rng(8)
    js = rng.permutation(np.flatnonzero(tol.evaluated_mask(n)))
    for b, cnt in enumerate((1047, 1048)):
        spec = np.zeros(n, np.complex128)
        spec[js[:cnt]] = n * np.exp(2j * np.pi * rng.uniform(size=cnt))
        x[b] = np.fft.ifft(spec).astype(np.complex64) * 0.01
    with Plan(n, FS, 1e9, max_batch=2, window_type=capi.WIN_RECTANGULAR) as plan:
        plan.submit_device(0, _to_dev(torch_cuda, x), 2)
        p, _, _ = plan.collect(0)
    thr = float(np.sort(p[0])[-1047:].min()) - 5.0        # far below the planted lines, far above the rest
    assert np.sort(p[0])[-1048] < thr - 20
    with Plan(n, FS, thr, max_batch=2, window_type=capi.WIN_RECTANGULAR, max_hits=4096) as plan:
        plan.submit_device(0, _to_dev(torch_cuda, x), 2)
        p, h, t = plan.collect(0, hit_cap=4096)
    assert len(h) == 1047 + 1048 and t.tolist() == [0, 1]


def test_hit_capacity_truncation(torch_cuda, oracle_mod):
    """More hits than the caller's buffer / the plan's pinned list: SCN_E_TRUNCATED with the FIRST records of the ordered
    list in the buffer, the true total in n_hits, and nothing lost -- scn_collect_more walks the rest."""
    n, nb = 4096, 4
    x = synth.cfloat_batch(n, nb, seed=9)
    fc = 100e6 + 6e6 * np.arange(nb)
    _, h_ref, _ = oracle_mod.Oracle(n, FS, -200.0).run(x, fc)
    assert len(h_ref) == nb * 3066
    with Plan(n, FS, -200.0, max_batch=nb, max_hits=1000) as plan:
        plan.submit_device(0, _to_dev(torch_cuda, x), nb, fc)
        with pytest.raises(capi.ScannerError) as e:
            plan.collect(0, hit_cap=5000)
        assert e.value.status == capi.E_TRUNCATED and plan.last_n_hits == len(h_ref)
        _assert_hits_equal(plan.hits_view(0), h_ref[:1000])                 # what the plan keeps in pinned memory
        _assert_hits_equal(plan.collect_more(0, 0, len(h_ref)), h_ref)      # ... and the whole list, from the GPU
        _assert_hits_equal(plan.collect_more(0, 2500, 777), h_ref[2500:3277])
        assert len(plan.collect_more(0, len(h_ref), 10)) == 0
        # the plan stays usable; the default collect fetches everything by itself
        plan.submit_device(0, _to_dev(torch_cuda, x), nb, fc)
        _, h, _ = plan.collect(0)
        _assert_hits_equal(h, h_ref)
        with pytest.raises(capi.ScannerError) as e2:                        # a submitted slot has no list to walk yet
            plan.submit_device(1, _to_dev(torch_cuda, x), nb, fc)
            plan.collect_more(1, 0, 10)
        assert e2.value.status == capi.E_STATE
        plan.collect(1)
    with Plan(n, FS, 1e9, max_batch=nb) as plan:           # and a quiet plan reports zero
        plan.submit_device(0, _to_dev(torch_cuda, x), nb)
        p, h, t = plan.collect(0)
        assert plan.last_n_hits == 0 and len(h) == 0 and len(plan.hits_view(0)) == 0


def test_long_hit_list_through_every_fetch_path(torch_cuda, oracle_mod):
    """100 buffers with every evaluated bin a hit = 306 600 records: the first submit's list is built on demand and copied at
    collect time, the second one's eagerly with a prefetch of the predicted size; then the zero-copy view and a window in
    the middle of the list."""
    n, nb = 4096, 100
    x = synth.cfloat_batch(n, nb, seed=77)
    fc = 100e6 + 6e6 * np.arange(nb)
    _, h_ref, _ = oracle_mod.Oracle(n, FS, -200.0).run(x, fc, threads=8)
    assert len(h_ref) == nb * 3066
    with Plan(n, FS, -200.0, max_batch=nb, max_hits=len(h_ref) + 10, flags=capi.OUT_HITS) as plan:
        for rounds in range(2):
            plan.submit_device(rounds, _to_dev(torch_cuda, x), nb, fc)
            _, h, _ = plan.collect(rounds, hit_cap=len(h_ref) + 10)
            _assert_hits_equal(h, h_ref)
        _assert_hits_equal(plan.hits_view(1), h_ref)
        _assert_hits_equal(plan.collect_more(1, (1 << 18) - 5, 10), h_ref[(1 << 18) - 5:(1 << 18) + 5])


def test_hit_list_eager_prefetch_and_top_up(torch_cuda, oracle_mod):
    """A caller that asks for records gets the list built beside the next launch and a DMA of the PREDICTED number of
    records (last total + 25 %); when a batch holds many more than predicted the rest is copied at collect time, and a
    counts-only collect in between switches the eager build off again.  Every list must equal the oracle's whatever path
    it took: quiet, loud (top-up), quiet (over-prediction), counts-only, then records again (built on demand)."""
    n, nb = 4096, 48
    quiet = synth.cfloat_batch(n, nb, seed=31, sigma=0.02, max_tones=1)
    loud = synth.cfloat_batch(n, nb, seed=32, sigma=0.3, max_tones=4)
    fc = 400e6 + 6e6 * np.arange(nb)
    o = oracle_mod.Oracle(n, FS, 9.0)
    refs = {id(quiet): o.run(quiet, fc)[1], id(loud): o.run(loud, fc)[1]}
    assert len(refs[id(loud)]) > 4 * len(refs[id(quiet)]) > 0
    dq, dl = _to_dev(torch_cuda, quiet), _to_dev(torch_cuda, loud)
    with Plan(n, FS, 9.0, max_batch=nb, max_hits=nb * n, flags=capi.OUT_HITS) as plan:
        seq = [(quiet, dq, True), (quiet, dq, True), (loud, dl, True), (loud, dl, True), (quiet, dq, True), (quiet, dq, False),
               (loud, dl, False), (loud, dl, True), (quiet, dq, True)]
        pend = []
        for k, (x, d, want) in enumerate(seq):
            if len(pend) == 2:
                k0, x0, want0 = pend.pop(0)
                _, h, t = plan.collect(k0 & 1, want_power=False, want_hits=want0)
                if want0:
                    _assert_hits_equal(h, refs[id(x0)])
                else:
                    assert plan.last_n_hits == len(refs[id(x0)])
            plan.submit_device(k & 1, d, nb, fc)
            pend.append((k, x, want))
        for k0, x0, want0 in pend:
            _, h, t = plan.collect(k0 & 1, want_power=False, want_hits=want0)
            if want0:
                _assert_hits_equal(h, refs[id(x0)])
        # an empty submit is a valid sweep result: no hits, an empty view
        plan.submit_device(0, dq, 0, [])
        _, h, t = plan.collect(0, want_power=False)
        assert len(h) == 0 and len(t) == 0 and len(plan.hits_view(0)) == 0


def test_negative_frequency_cast_follows_x86(torch_cuda, oracle_mod):
    """A sweep that starts at 0 Hz has start_frequency = 3 MHz - 4 MHz < 0 for its first centre; process.cpp:57 casts the
    (negative) double of the lowest evaluated bins to uint64, which on the reference's x86-64 build wraps -- the
    compaction kernel reproduces that, bit for bit with the oracle compiled here."""
    n = 4096
    centres, i0 = np.array([0]), np.array([515])
    x = synth.c4_shard(n, 0, 2, centres, i0, seed=3)
    fc = capi.frequency_table(FS, 0.0, 2 * 0.75 * FS)[1]
    (p, h, t), (p_ref, h_ref, t_ref) = _run_both(torch_cuda, oracle_mod, n, capi.KIND_FLOAT_COMPLEX, x, fc, None, 10.0)
    assert h_ref["i"][0] == 512 and h_ref["freq_hz"][0] == np.uint64(2**64 - 64)
    _assert_hits_equal(h, h_ref)


def test_spectrum_only_and_hits_only_modes(torch_cuda, oracle_mod):
    n, nb = 4096, 16
    x = synth.cfloat_batch(n, nb, seed=12)
    thr = _clear_threshold(oracle_mod, n, [x], 9.5)
    o = oracle_mod.Oracle(n, FS, thr)
    p_ref, h_ref, t_ref = o.run(x, np.full(nb, 1e9))
    assert len(h_ref) > 50
    d = _to_dev(torch_cuda, x)
    with Plan(n, FS, thr, max_batch=nb, flags=capi.OUT_SPECTRUM) as plan:
        plan.submit_device(0, d, nb, np.full(nb, 1e9))
        p, h, t = plan.collect(0)
        assert h is None and t is None
        tol.compare_spectra(p, p_ref)
    with Plan(n, FS, thr, max_batch=nb, flags=capi.OUT_HITS, max_hits=1 << 16) as plan:
        plan.submit_device(0, d, nb, np.full(nb, 1e9))
        p, h, t = plan.collect(0, hit_cap=1 << 16)
        assert p is None
        _assert_hits_equal(h, h_ref)
        assert np.array_equal(t, t_ref)
    # caller-provided device destination for the spectra
    out = torch_cuda.empty((nb, n), dtype=torch_cuda.float32, device="cuda")
    with Plan(n, FS, thr, max_batch=nb) as plan:
        plan.submit_device(1, d, nb, np.full(nb, 1e9), d_power_db=out)
        plan.wait(1)
        p2, _, _ = plan.collect(1)
    assert np.array_equal(out.cpu().numpy(), p2)
    tol.compare_spectra(p2, p_ref)


# ---------------------------------------------------------------------------------------
# ragged / edge batches
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("nb", [0, 1, 2, 255, 1025])
def test_ragged_batches(torch_cuda, oracle_mod, nb):
    n = 4096
    x = synth.cfloat_batch(n, max(nb, 1), seed=30 + nb, max_tones=1)[:nb]
    fc = 1e9 + 6e6 * np.arange(nb)
    thr = _clear_threshold(oracle_mod, n, [x], 12.0)
    with Plan(n, FS, thr, max_batch=1100, max_hits=1 << 16) as plan:
        d = _to_dev(torch_cuda, x) if nb else torch_cuda.empty(8, dtype=torch_cuda.uint8, device="cuda")
        plan.submit_device(0, d, nb, fc)
        p, h, t = plan.collect(0, hit_cap=1 << 16)
    assert p.shape == (nb, n) and t.shape == (nb,)
    if nb:
        p_ref, h_ref, t_ref = oracle_mod.Oracle(n, FS, thr).run(x, fc, threads=4)
        tol.compare_spectra(p, p_ref)
        _assert_hits_equal(h, h_ref)
        assert np.array_equal(t, t_ref)
    with Plan(n, FS, 12.0, max_batch=4) as plan:
        with pytest.raises(capi.ScannerError) as e:
            plan.submit_device(0, _to_dev(torch_cuda, synth.cfloat_batch(n, 5, 1)), 5)
        assert e.value.status == capi.E_INVALID


def test_counts_by_kernel_stores_and_by_dma_agree(torch_cuda, oracle_mod):
    """The per-buffer counts reach the host by the kernel's own stores to pinned memory for launches of up to 4096 buffers
    (and whenever the ordered list follows eagerly) and by a DMA above (scn_api.hip, submit_common): both sides of that
    line, on one plan, counts-only collects first (no eager list), then with records (eager list: stores again)."""
    n, nb = 1024, 4200
    x = synth.cfloat_batch(n, nb, seed=77, max_tones=2)
    fc = 2e9 + 6e6 * np.arange(nb)
    thr = _clear_threshold(oracle_mod, n, [x], 9.0)
    o = oracle_mod.Oracle(n, FS, thr)
    o.params.trigger_count = 3
    _, h_ref, t_ref = o.run(x, fc, want_power=False, threads=8)
    c_ref = np.bincount(h_ref["seq_id"].astype(np.int64), minlength=nb)
    assert t_ref.any() and not t_ref.all()
    d = _to_dev(torch_cuda, x)
    with Plan(n, FS, thr, max_batch=nb, max_hits=len(h_ref) + 64, flags=capi.OUT_HITS, trigger_count=3) as plan:
        for count in (4096, 4097, nb, 4096):        # stores, DMA, DMA, stores
            plan.submit_device(0, d, count, fc[:count])
            _, h, t = plan.collect(0, want_hits=False)
            assert h is None and np.array_equal(t, t_ref[:count]), count
        for count in (nb, 4096, nb):                 # with records: the first builds its list on demand, the next two eagerly
            plan.submit_device(1, d, count, fc[:count])
            _, h, t = plan.collect(1, hit_cap=len(h_ref) + 64)
            _assert_hits_equal(h, h_ref[h_ref["seq_id"] < count])
            assert np.array_equal(t, t_ref[:count])
            assert np.array_equal(np.bincount(h["seq_id"].astype(np.int64), minlength=count), c_ref[:count])


@pytest.mark.parametrize("n", [1024, 3000, 4096, 8192, 16384, 65536])
def test_hip_against_rocfft_through_torch(torch_cuda, n):
    """A third opinion on the transform, on the same GPU: torch.fft.fft on complex64 runs the vendor's FFT library (rocFFT / hipFFT) --
    another independent production float32 FFT, never on the product path.  Both it and the HIP kernels sit a few 1e-6 from the
    float64 spectrum and within the 1e-5 bar of each other (the plan's own window table multiplies both)."""
    torch = torch_cuda
    nb = 12 if n < 65536 else 4
    x = synth.cfloat_batch(n, nb, seed=77 + n)
    with Plan(n, FS, 1e9, max_batch=nb) as plan:
        w = plan.window()
        plan.submit_device(0, _to_dev(torch, x), nb)
        p, _, _ = plan.collect(0)
    xw = torch.from_numpy(x).cuda() * torch.from_numpy(w).cuda()                 # float32 multiply, as VOLK's kernel does
    X = torch.fft.fft(xw, dim=-1)                                                # complex64: the vendor library's float32 transform
    P = (X.real.double() ** 2 + X.imag.double() ** 2).cpu().numpy()
    with np.errstate(divide="ignore"):
        db_vendor = 5.0 * np.log10(P)
    X64 = np.fft.fft(x.astype(np.complex128) * w.astype(np.float64), axis=-1)
    with np.errstate(divide="ignore"):
        db64 = 5.0 * np.log10(X64.real ** 2 + X64.imag ** 2)
    print(n, "HIP vs rocFFT float32:", tol.compare_spectra(p, db_vendor)["max_rel_power_vs_max_bin_mean"],
          "| rocFFT vs float64:", tol.compare_spectra(db_vendor, db64)["max_rel_power_vs_max_bin_mean"],
          "| HIP vs float64:", tol.compare_spectra(p, db64)["max_rel_power_vs_max_bin_mean"])


@pytest.mark.parametrize("wt", ["HAMMING", "HANN", "BLACKMAN", "RECTANGULAR", "KAISER", "BLACKMAN_HARRIS", "BARTLETT", "FLATTOP"])
def test_every_window_type(torch_cuda, oracle_mod, wt):
    """process.cpp:18 hands any gr::fft::window::win_type to window::build: the plan's table has the oracle's bits for each, and the
    spectra + hits follow (a 1024-point fused kernel, the 3000-point mixed-radix one and the four-step pair take the table the same way)."""
    o_t, p_t = getattr(oracle_mod, "WIN_" + wt), getattr(capi, "WIN_" + wt)
    for n, nb in ((1024, 16), (3000, 6), (32768, 3)):
        x = synth.cfloat_batch(n, nb, seed=n + o_t)
        fc = 3e6 + 6e6 * np.arange(nb)
        with oracle_mod.window_type(o_t):
            p_ref, _, _ = oracle_mod.Oracle(n, FS, 1e9).run(x, threads=4)
            thr = tol.pick_threshold(p_ref, n, start=6.0)
            p_ref, h_ref, t_ref = oracle_mod.Oracle(n, FS, thr).run(x, fc, threads=4)
        with Plan(n, FS, thr, max_batch=nb, max_hits=nb * n, window_type=p_t) as plan:
            assert np.array_equal(plan.window(), oracle_mod.window(o_t, n))
            plan.submit_device(0, _to_dev(torch_cuda, x), nb, fc)
            p, h, t = plan.collect(0, hit_cap=nb * n)
        tol.compare_spectra(p, p_ref)
        _assert_hits_equal(h, h_ref)
        assert np.array_equal(t, t_ref)
    with pytest.raises(capi.ScannerError):
        Plan(1024, FS, 10.0, window_type=9)


def test_total_and_trigger_bitmap_from_the_gpu(torch_cuda, oracle_mod):
    """Launches of 2^18 buffers and more (16 ... 128-point plans at the bench's batch): the host gets the batch's total and ONE BIT per
    buffer (process_fft's return value, hits > trigger_count, process.cpp:62) from a reduction on the GPU instead of 4 bytes per
    buffer over PCIe; the counts stay on the device, where the list kernels rank the records from them.  Both sides of that line
    on one plan, a count that is not a multiple of 32 or of 4, trigger flags against the oracle's, records intact."""
    n, nb = 16, (1 << 18) + 37
    rng = np.random.default_rng(5)
    x = (rng.standard_normal((nb, n, 2), dtype=np.float32) * np.float32(0.3)).view(np.complex64).reshape(nb, n)
    fc = 1e9 + 6e6 * (np.arange(nb) % 1000)
    thr = -3.0
    o = oracle_mod.Oracle(n, FS, thr)
    o.params.trigger_count = 2
    p_ref, h_ref, t_ref = o.run(x, fc, threads=8)
    ok = ~(tol.flip_unsafe(p_ref, thr) & tol.evaluated_mask(n)[None, :]).any(axis=1)      # buffers whose every bin is clear of the threshold
    assert t_ref.any() and not t_ref.all() and ok.mean() > 0.9
    d = _to_dev(torch_cuda, x)
    with Plan(n, FS, thr, max_batch=nb, max_hits=len(h_ref) + 1024, flags=capi.OUT_HITS, trigger_count=2) as plan:
        for count in (nb, (1 << 18) - 1, 1 << 18, nb - 2):        # bitmap, counts by DMA, bitmap, bitmap (a partial quad, a partial word)
            plan.submit_device(0, d, count, fc[:count])
            total = plan.collect_counts(0)
            plan.submit_device(0, d, count, fc[:count])
            _, h, t = plan.collect(0, want_hits=False)
            assert np.array_equal(t[ok[:count]], t_ref[:count][ok[:count]]), count
            assert total == plan.last_n_hits
            assert abs(total - int((h_ref["seq_id"] < count).sum())) <= int((~ok[:count]).sum()) * n
        plan.submit_device(1, d, nb, fc)                           # ... and the records behind a bitmap collect are the oracle's
        _, h, t = plan.collect(1, hit_cap=len(h_ref) + 1024)
        keep_ref, keep = ok[h_ref["seq_id"].astype(np.int64)], ok[h["seq_id"].astype(np.int64)]
        _assert_hits_equal(h[keep], h_ref[keep_ref])
        assert np.array_equal(t[ok], t_ref[ok])


def test_special_inputs(torch_cuda, oracle_mod):
    """all-zero buffer (-inf everywhere, no hits), a unit impulse (flat spectrum), full-scale DC."""
    n = 4096
    x = np.zeros((3, n), np.complex64)
    x[1, 0] = 1.0
    x[2, :] = 1.0
    with Plan(n, FS, -50.0, max_batch=3, window_type=capi.WIN_RECTANGULAR, max_hits=1 << 15) as plan:
        plan.submit_device(0, _to_dev(torch_cuda, x), 3)
        p, h, t = plan.collect(0, hit_cap=1 << 15)
    assert np.all(np.isneginf(p[0])) and not np.any(h["seq_id"] == 0)
    assert np.abs(p[1]).max() < 1e-5                         # |X| = 1 everywhere -> 0 dB
    assert abs(p[2, 0] - 5 * np.log10(float(n) ** 2)) < 1e-4 and np.all(p[2, 1:] < -50)


# ---------------------------------------------------------------------------------------
# full BASELINE size (C2: batch 8192 x 4096): every buffer against the oracle + properties
# ---------------------------------------------------------------------------------------
def test_c2_full_size(torch_cuda, oracle_mod):
    n, nb = 4096, 8192
    xd = synth.cfloat_batch_torch(n, nb, seed=2, device="cuda")
    x = xd.cpu().numpy().view(np.complex64).reshape(nb, n)
    fc = 3e6 + 6e6 * np.arange(nb)
    o = oracle_mod.Oracle(n, FS, 10.0)
    p_ref, h_ref, t_ref = o.run(x, fc, threads=8)
    with Plan(n, FS, 10.0, max_batch=nb, max_hits=1 << 22) as plan:
        plan.submit_device(0, xd, nb, fc)
        p, h, t = plan.collect(0, hit_cap=1 << 22)
        # idempotence: a second launch over the same input gives the same bits
        plan.submit_device(1, xd, nb, fc)
        p2, h2, _ = plan.collect(1, hit_cap=1 << 22)
    assert np.array_equal(p, p2) and np.array_equal(h, h2)
    fig = tol.compare_spectra(p, p_ref)
    print("C2 full size vs oracle:", fig)
    # hits: exact wherever the oracle's value is outside the guard band
    m = tol.evaluated_mask(n)
    near = np.zeros((nb, n), bool)
    near[:, m] = np.abs(p_ref[:, m] - 10.0) < tol.GUARD_DB
    print("guard-band population:", int(near.sum()), "of", nb * int(m.sum()))
    key = lambda a: a["seq_id"].astype(np.int64) * n + a["i"]       # noqa: E731
    jj = lambda a: (a["i"].astype(np.int64) + n // 2) % n           # noqa: E731
    keep_ref = ~near[h_ref["seq_id"].astype(np.int64), jj(h_ref)]
    keep = ~near[h["seq_id"].astype(np.int64), jj(h)]
    assert np.array_equal(key(h[keep]), key(h_ref[keep_ref]))
    assert np.array_equal(h[keep]["freq_hz"], h_ref[keep_ref]["freq_hz"])
    assert np.array_equal(t, t_ref)
    # Parseval (size-independent property): sum_k |X_k|^2 = N * sum_n |x_n w_n|^2
    w = o.window().astype(np.float64)
    lhs = tol.db_to_power(p.astype(np.float64)).sum(axis=1)
    rhs = n * ((np.abs(x.astype(np.complex128)) ** 2) * w ** 2).sum(axis=1)
    assert np.abs(lhs / rhs - 1).max() < 2e-6


# ---------------------------------------------------------------------------------------
# time-domain mode (process.cpp:203-237), the CLI's default mode (scan.cpp:87)
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,kind,enob,dc", [
    (8192, capi.KIND_FLOAT_COMPLEX, 12, False),
    (1000, capi.KIND_FLOAT_COMPLEX, 12, False),      # the time-domain path takes any sample count
    (8192, capi.KIND_SHORT_COMPLEX, 12, True),
    (4096, capi.KIND_SHORT, 12, False),
    (2048, capi.KIND_BYTE_COMPLEX, 8, True),
    # the streaming form (one wave per buffer, 16-byte loads): smallest buffers (half a load iteration), planar
    # with DC, every format at 1024; and a sample count that is not a multiple of 8 (the per-sample form)
    (1024, capi.KIND_BYTE_COMPLEX, 8, False),
    (1024, capi.KIND_SHORT, 12, True),
    (1024, capi.KIND_SHORT_COMPLEX, 14, False),
    (8192, capi.KIND_SHORT, 12, True),
    (8192, capi.KIND_BYTE_COMPLEX, 8, False),
    (2048, capi.KIND_FLOAT_COMPLEX, 12, False),
    (1004, capi.KIND_SHORT_COMPLEX, 12, True),
])
def test_time_domain_mode(torch_cuda, oracle_mod, n, kind, enob, dc):
    nb = 50
    rng = np.random.default_rng(n)
    x = (rng.standard_normal((nb, n)) + 1j * rng.standard_normal((nb, n))).astype(np.complex64) * 0.05
    x[::3, 17] += 0.9                                  # a strong sample in every third buffer
    x[5] = 0                                           # an all-zero buffer: -inf dB everywhere
    raw = synth.quantize(x, kind)
    o = oracle_mod.Oracle(n, FS, 0.0, kind=kind, enob=enob, correct_dc=dc)
    flat = raw.reshape(nb, -1)
    ref = [o.time_domain(o.convert(flat[b]), threshold=-1.5) for b in range(nb)]
    ref_max = np.array([r[1] for r in ref], np.float32)
    ref_min = np.array([r[2] for r in ref], np.float32)
    ref_hit = np.array([r[0] for r in ref], np.uint8)
    with Plan(n, FS, -1.5, kind=kind, enob=enob, correct_dc=dc, max_batch=64, mode=capi.MODE_TIME_DOMAIN) as plan:
        plan.submit_device(0, _to_dev(torch_cuda, raw), nb, np.arange(nb) * 1e6)
        mx, mn, ab = plan.collect_time_domain(0)
        with pytest.raises(capi.ScannerError):
            plan.submit_device(0, _to_dev(torch_cuda, raw), nb, np.arange(nb) * 1e6)
            plan.collect(0)                            # wrong collect for this mode
    fin_max, fin_min = np.isfinite(ref_max), np.isfinite(ref_min)   # (8-bit noise has an exactly zero sample in every buffer)
    assert np.array_equal(np.isneginf(mn), np.isneginf(ref_min)) and np.array_equal(np.isfinite(mx), fin_max)
    assert np.abs(mx[fin_max] - ref_max[fin_max]).max(initial=0) < 1e-4
    assert np.abs(mn[fin_min] - ref_min[fin_min]).max(initial=0) < 2e-3
    clear = np.abs(ref_max - (-1.5)) > 1e-3
    assert np.array_equal(ab[clear], ref_hit[clear]) and ref_hit.sum() >= nb // 3
    # the all-zero buffer keeps the reference's odd initial maximum (process.cpp:207)
    assert mx[5] == np.float32(1.17549435e-38) == ref_max[5]


@pytest.mark.parametrize("n,kind", [(1024, capi.KIND_FLOAT_COMPLEX), (128, capi.KIND_SHORT_COMPLEX), (1000, capi.KIND_FLOAT_COMPLEX)])
def test_indexed_submit_gives_the_same_records(torch_cuda, oracle_mod, n, kind):
    """scn_plan_set_table + scn_submit*_indexed: buffer b of a launch carries entry (first + b) % count of the plan's GPU-resident
    frequency table -- what GetNextFrequency hands the source retune by retune, wrapping (frequencyTable.cpp:38-46) -- and the
    records are those of the same launch with the centres passed per buffer, byte for byte, and the oracle's; with and
    without sequence ids, from device memory and from the pinned slot."""
    nb, count, first = 300, 37, 11
    table = 1e9 + 6e6 * np.arange(count) + 0.25
    fc = table[(first + np.arange(nb)) % count]
    x = synth.cfloat_batch(n, nb, seed=77, sigma=0.05)
    raw = synth.quantize(x, kind)
    enob = 12
    thr = tol.pick_threshold(oracle_mod.Oracle(n, FS, 1e9, kind=kind, enob=enob).run(raw)[0], n, start=12.0)
    seq = np.arange(nb, dtype=np.uint64) * 3 + 5
    _, ref_seq, _ = oracle_mod.Oracle(n, FS, thr, kind=kind, enob=enob).run(raw, fc, seq)
    _, ref_idx, _ = oracle_mod.Oracle(n, FS, thr, kind=kind, enob=enob).run(raw, fc, np.arange(nb, dtype=np.uint64))
    assert len(ref_seq) > 50 and len(np.unique(ref_seq["freq_hz"] // 6000000)) > 20
    d = _to_dev(torch_cuda, raw)
    with Plan(n, FS, thr, kind=kind, enob=enob, max_batch=nb, max_hits=1 << 16) as plan:
        with pytest.raises(capi.ScannerError) as e:   # no table yet
            plan.submit_device(0, d, nb, first_index=0)
        assert e.value.status == capi.E_STATE
        plan.set_table(table)
        with pytest.raises(capi.ScannerError) as e:   # outside the table
            plan.submit_device(0, d, nb, first_index=count)
        assert e.value.status == capi.E_INVALID
        for ids, ref in ((seq, ref_seq), (None, ref_idx)):
            plan.submit_device(0, d, nb, fc, ids)
            _, explicit, _ = plan.collect(0, want_power=False, hit_cap=1 << 16)
            plan.submit_device(1, d, nb, seq_ids=ids, first_index=first)
            _, indexed, _ = plan.collect(1, want_power=False, hit_cap=1 << 16)
            assert indexed.tobytes() == explicit.tobytes()
            _assert_hits_equal(indexed, ref)
            hb = plan.host_buffer(2)
            hb[: raw.nbytes] = np.ascontiguousarray(raw).view(np.uint8).reshape(-1)
            plan.submit(2, nb, seq_ids=ids, first_index=first)
            _, staged, _ = plan.collect(2, want_power=False, hit_cap=1 << 16)
            assert staged.tobytes() == explicit.tobytes()
        # an explicit submit after an indexed one on the same slot, and the other way round (the two generations of a slot)
        plan.submit_device(0, d, nb, seq_ids=seq, first_index=first)
        plan.collect(0, want_power=False, want_hits=False)
        plan.submit_device(0, d, nb, fc + 1e6, seq)
        _, shifted, _ = plan.collect(0, want_power=False, hit_cap=1 << 16)
        assert np.array_equal(shifted["freq_hz"], ref_seq["freq_hz"] + 1000000)
        # a pending submit reads the table: it cannot be replaced under it
        plan.submit_device(0, d, nb, seq_ids=seq, first_index=first)
        with pytest.raises(capi.ScannerError) as e:
            plan.set_table(table + 1.0)
        assert e.value.status == capi.E_STATE
        plan.collect(0, want_power=False, want_hits=False)
        plan.set_table(table[:5] + 2e6)
        plan.submit_device(0, d, nb, seq_ids=seq, first_index=4)
        _, small, _ = plan.collect(0, want_power=False, hit_cap=1 << 16)
        _, ref_small, _ = oracle_mod.Oracle(n, FS, thr, kind=kind, enob=enob).run(raw, (table[:5] + 2e6)[(4 + np.arange(nb)) % 5], seq)
        _assert_hits_equal(small, ref_small)


@pytest.mark.parametrize("n,nb", [(16, (1 << 19) + 5), (32, (1 << 19) + 8), (64, 40000)])
def test_total_of_a_many_buffer_launch_without_the_walk(torch_cuda, oracle_mod, n, nb):
    """From 2^19 buffers per launch the batch's total is summed on the GPU and scn_collect without trigger flags reads one word
    instead of walking the counts (scn_hit_total_kernel): the same number as the walk's (scn_collect with trigger flags; every
    collect of the smaller launch), as the number of records, and as the oracle's, launch after launch on the same slot (the
    kernel's two device words return to zero)."""
    rng = np.random.default_rng(5)
    x = (rng.standard_normal((nb, n, 2), dtype=np.float32) * np.float32(0.05)).view(np.complex64).reshape(nb, n)
    # a threshold far enough into the noise's own tail that its guard band is empty over ALL the launch's bins: a few hundred hits
    p_ref = oracle_mod.Oracle(n, FS, 1e9).run(x, threads=8)[0]
    vals = p_ref[:, tol.evaluated_mask(n, 0.75, 4)]
    thr = tol.pick_threshold(p_ref, n, start=float(np.partition(vals.ravel(), -3000)[-3000]))
    _, ref, _ = oracle_mod.Oracle(n, FS, thr).run(x, threads=8)
    assert len(ref) > 100
    d = _to_dev(torch_cuda, x)
    with Plan(n, FS, thr, max_batch=nb, max_hits=max(len(ref), 1) + 64) as plan:
        plan.set_table(np.full(1, 1e9))
        for rep in range(3):
            for slot in (0, 1):
                plan.submit_device(slot, d, nb, first_index=0)
            for slot in (0, 1):
                assert plan.collect_counts(slot) == len(ref)       # no trigger flags: the word the GPU left
        part = nb - 3                                  # another count on the same slot (not a multiple of four; still on the GPU-total side) ...
        _, ref_part, _ = oracle_mod.Oracle(n, FS, thr).run(x[:part], threads=8)
        with Plan(n, FS, thr, max_batch=part, max_hits=len(ref) + 64) as small:   # ... and in a plan of its own, whatever the first one left behind
            small.set_table(np.full(1, 1e9))
            small.submit_device(0, d[: part * n * 8], part, first_index=0)
            assert small.collect_counts(0) == len(ref_part)
        plan.submit_device(0, d[: part * n * 8], part, first_index=0)
        assert plan.collect_counts(0) == len(ref_part)
        keep = nb // 3
        plan.submit_device(0, d[: keep * n * 8], keep, first_index=0)         # (below 2^19 buffers: the walk)
        assert plan.collect_counts(0) == len(oracle_mod.Oracle(n, FS, thr).run(x[:keep], threads=8)[1]) < len(ref)
        plan.submit_device(0, d, nb, first_index=0)
        _, hits, trig = plan.collect(0, want_power=False)             # trigger flags: the walk
        assert plan.last_n_hits == len(ref) == len(hits) and trig.shape == (nb,)

"""The in-tree arithmetic of the hot path, pinned against the REFERENCE ITSELF: utility.cpp -- the three integer ->
complex-float converters (utility.cpp:9-84) and the dB map (utility.cpp:86-98, row a6) --
compiles from /root/reference with `make -C oracle ref` (oracle/ref_dsp_binding.cpp says how its one foreign include,
<fftw3.h> for the type fftwf_complex, is satisfied from this image's hipFFTW header).
tests/golden/utility_ref.npz holds inputs and that build's outputs (tests/golden/make_utility_ref.py); held to it,
bit for bit: the oracle's restatement (always, on CPU), the live reference object on random inputs (where oracle/_ref is
present), and -- on the GPU -- the PRODUCT's own K1 (scn_convert_raw, the load phase of the fused kernels).
Still unpinned by reference code: the window (GNU Radio), the multiply (VOLK), the FFT (FFTW) and process_fft
(process.cpp includes all three)."""
import os

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = np.load(os.path.join(ROOT, "tests", "golden", "utility_ref.npz"))
NAMES = [str(x) for x in GOLD["names"]]


@pytest.mark.parametrize("name", NAMES)
def test_oracle_converters_match_reference_fixture(oracle_mod, name):
    kind, n, enob, dc = (int(v) for v in GOLD[name + "_meta"])
    got = oracle_mod.Oracle(n, kind=kind, enob=enob, correct_dc=bool(dc)).convert(GOLD[name + "_in"])
    assert got.tobytes() == GOLD[name + "_out"].tobytes(), name       # every float bit for bit, quirks included


def test_reference_quirks_are_in_the_fixture():
    """What SURVEY.md 8a found by running the reference object, now as data from the reference object."""
    e16 = GOLD["s16c_extremes_e16_dc0_out"]
    assert e16[0].real == np.float32(32767 / -32768.0) and e16[0].imag == np.float32(1.0)      # scale sign flips at ENOB 16
    e8 = GOLD["s8c_n64_e8_dc0_out"]
    src = GOLD["s8c_n64_e8_dc0_in"]
    assert np.array_equal(e8.real, (src[:, 0].astype(np.float32) / np.float32(-128.0)).astype(np.float32))  # ... and at 8
    neg = GOLD["s16c_negmean_dc1_out"]
    assert np.abs(neg.real).min() > 1000.0                        # int32 /= uint32: a negative sum becomes a huge "mean"


def test_oracle_db_map_matches_reference_fixture(oracle_mod):
    X = GOLD["mag_in"]
    with np.errstate(all="ignore"):
        got = oracle_mod.Oracle(X.size).magnitude(X)
    want = GOLD["mag_out"]
    assert np.array_equal(np.isneginf(got), np.isneginf(want)) and np.isneginf(want[0])        # zero bin -> -inf
    assert got.tobytes() == want.tobytes()                         # the double-log2 build of utility.cpp:86-98


needs_ref = pytest.mark.skipif(not O.ref_dsp_available(), reason="oracle/_ref/libref_utility.so not built (needs /root/reference)")


@needs_ref
@settings(max_examples=200, deadline=None)
@given(kind=st.sampled_from([O.KIND_SHORT_COMPLEX, O.KIND_SHORT, O.KIND_BYTE_COMPLEX]), n=st.integers(1, 700),
       enob=st.integers(2, 16), dc=st.booleans(), seed=st.integers(0, 2**31 - 1), offset=st.integers(-3000, 3000))
def test_live_reference_converters_on_random_buffers(oracle_mod, kind, n, enob, dc, seed, offset):
    rng = np.random.default_rng(seed)
    if kind == O.KIND_BYTE_COMPLEX:
        enob = min(enob, 8)
        raw = np.clip(rng.integers(-128, 128, size=(n, 2)) + offset // 32, -128, 127).astype(np.int8)
    else:
        raw = np.clip(rng.integers(-2048, 2048, size=(n, 2)) + offset, -32768, 32767).astype(np.int16)
        if kind == O.KIND_SHORT:
            raw = np.ascontiguousarray(raw.T)
    want = oracle_mod.ref_convert(kind, raw, n, enob, dc)
    got = oracle_mod.Oracle(n, kind=kind, enob=enob, correct_dc=dc).convert(raw)
    assert got.tobytes() == want.tobytes()


@needs_ref
@settings(max_examples=60, deadline=None)
@given(seed=st.integers(0, 2**31 - 1), n=st.integers(1, 3000), exp=st.integers(-18, 17))
def test_live_reference_db_map_on_random_spectra(oracle_mod, seed, n, exp):
    rng = np.random.default_rng(seed)
    X = ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 10.0 ** exp).astype(np.complex64)
    with np.errstate(all="ignore"):
        assert oracle_mod.Oracle(n).magnitude(X).tobytes() == oracle_mod.ref_magnitude(X).tobytes()


@pytest.mark.gpu
def test_product_k1_matches_reference_fixture(built_lib):
    """The HIP path's own convert (scn_convert_raw = the RawLoader the fused kernels use) against the reference's output."""
    import torch

    from scanner_amd import Plan, capi

    assert torch.cuda.is_available()
    for name in NAMES:
        kind, n, enob, dc = (int(v) for v in GOLD[name + "_meta"])
        with Plan(n, kind=kind, enob=enob, correct_dc=bool(dc), max_batch=1, mode=capi.MODE_TIME_DOMAIN) as plan:
            got = plan.convert_raw(GOLD[name + "_in"])[0]
        assert got.tobytes() == GOLD[name + "_out"].tobytes(), name


@pytest.mark.gpu
@needs_ref
def test_product_k1_matches_the_live_reference_object(built_lib):
    """oracle/_ref travels to the GPU box (git-ignored, not gpurun-ignored): the HIP path's own K1 -- scn_convert_raw, the load
    phase of the fused kernels -- against the REFERENCE's compiled utility.cpp on random buffers there, bit for bit: every wire
    format, ENOB 2 .. 16, with and without DC removal, negative means (the int32 /= uint32 quirk) included."""
    import torch

    from scanner_amd import Plan, capi

    assert torch.cuda.is_available()
    rng = np.random.default_rng(20261004)
    for case in range(60):
        kind = int(rng.choice([O.KIND_SHORT_COMPLEX, O.KIND_SHORT, O.KIND_BYTE_COMPLEX]))
        n = int(rng.choice([16, 64, 700, 1024, 4096, 8192]))
        enob = int(rng.integers(2, 9 if kind == O.KIND_BYTE_COMPLEX else 17))
        dc = bool(rng.integers(0, 2))
        offset = int(rng.integers(-3000, 3000))
        if kind == O.KIND_BYTE_COMPLEX:
            raw = np.clip(rng.integers(-128, 128, size=(n, 2)) + offset // 32, -128, 127).astype(np.int8)
        else:
            raw = np.clip(rng.integers(-2048, 2048, size=(n, 2)) + offset, -32768, 32767).astype(np.int16)
            if kind == O.KIND_SHORT:
                raw = np.ascontiguousarray(raw.T)
        want = O.ref_convert(kind, raw, n, enob, dc)
        with Plan(n, kind=kind, enob=enob, correct_dc=dc, max_batch=1, mode=capi.MODE_TIME_DOMAIN) as plan:
            got = plan.convert_raw(raw)[0]
        assert got.tobytes() == want.tobytes(), (case, kind, n, enob, dc, offset)

"""the in-band header of HackRF sweep-mode transfers
(HackRFSource::interpolateSamples, hackRFSource.cpp:186-222).

Host-only arithmetic, so the C-ABI entry point (scn_hackrf_sweep_fixup) is compared with the oracle's
statement-by-statement restatement here on the CPU; the hand-derived known answers pin both."""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from oracle import oracle
from scanner_amd import capi

BLOCK_BYTES = 2 * 8192


def make_transfer(rng, n_blocks, freqs_hz, tag=True):
    """n_blocks 8192-sample int8 IQ blocks, each starting with the firmware's header."""
    t = rng.integers(0, 256, n_blocks * BLOCK_BYTES, dtype=np.uint8)
    for b in range(n_blocks):
        o = b * BLOCK_BYTES
        if tag:
            t[o] = t[o + 1] = 0x7F
            t[o + 2:o + 10] = np.frombuffer(np.uint64(freqs_hz[b % len(freqs_hz)]).tobytes(), np.uint8)
    return t


def both(t, offset):
    a = t.copy()
    fc, mism = capi.hackrf_sweep_fixup(a, offset)
    b, fc_o, mism_o = oracle.hackrf_interpolate(t, offset)
    assert fc == fc_o and mism == mism_o
    assert np.array_equal(a, b)
    return a, fc, mism


def test_known_answer_header_parsed_and_samples_patched(built_lib):
    rng = np.random.default_rng(5)
    f = 2_412_000_000
    t = make_transfer(rng, 16, [f])
    t[10], t[11] = 0x12, 0xF0                      # sample 5 = (18, -16)
    out, fc, mism = both(t, 7_500_000)
    assert fc == float(f + 7_500_000) and mism == 0
    # hand-derived: samples 0..4 become sample 5, nothing else changes
    assert out[:10].tolist() == [0x12, 0xF0] * 5
    assert np.array_equal(out[10:], t[10:])


def test_only_the_first_block_is_examined(built_lib):
    """The reference's block loop re-reads the head of the transfer (hackRFSource.cpp:192), so the
    headers of blocks 1.. stay in the sample stream and a different frequency there goes unnoticed."""
    rng = np.random.default_rng(6)
    t = make_transfer(rng, 4, [100_000_000, 200_000_000, 300_000_000, 400_000_000])
    t[10], t[11] = 1, 2
    out, fc, mism = both(t, 0)
    assert fc == 100_000_000.0 and mism == 0
    for b in range(1, 4):
        o = b * BLOCK_BYTES
        assert out[o] == 0x7F and out[o + 1] == 0x7F and np.array_equal(out[o:o + 12], t[o:o + 12])


def test_untagged_transfer_is_left_alone(built_lib):
    rng = np.random.default_rng(7)
    t = make_transfer(rng, 2, [0], tag=False)
    t[0] = 0x7F
    t[1] = 0x7E
    out, fc, mism = both(t, 123)
    assert fc == 123.0 and mism == 0 and np.array_equal(out, t)


def test_patch_value_that_looks_like_a_header(built_lib):
    """Sample 5 == (0x7F, 0x7F): after the first pass the head carries the tag again, so the second
    pass re-reads the 'frequency' from the patched bytes (a mismatch the reference reports on stdout)
    and averages the patch with the sample before block 1 -- hand-derived below."""
    rng = np.random.default_rng(8)
    t = make_transfer(rng, 2, [915_000_000])
    t[10] = t[11] = 0x7F
    t[BLOCK_BYTES - 2], t[BLOCK_BYTES - 1] = 0x05, 0xFB     # sample 8191 = (5, -5)
    out, fc, mism = both(t, 0)
    assert mism == 1
    assert fc == float(0x7F7F7F7F7F7F7F7F)
    # pass 2: post = ((127 + 5) / 2, (127 - 5) / 2) = (66, 61)
    assert out[:10].tolist() == [66, 61] * 5


def test_rejects_bad_arguments(built_lib):
    import ctypes as C
    L = capi.lib()
    fc = C.c_double()
    buf = (C.c_uint8 * 16)()
    assert L.scn_hackrf_sweep_fixup(None, 16, 0, C.byref(fc), None) == capi.E_INVALID
    assert L.scn_hackrf_sweep_fixup(buf, 16, 0, None, None) == capi.E_INVALID
    assert L.scn_hackrf_sweep_fixup(buf, 8, 0, C.byref(fc), None) == capi.E_INVALID
    assert L.scn_hackrf_sweep_fixup(buf, 16, 0, C.byref(fc), None) == capi.OK and fc.value == 0.0


@settings(max_examples=200, deadline=None)
@given(seed=st.integers(0, 2**32 - 1), n_blocks=st.integers(1, 5), tail=st.integers(0, 40),
       offset=st.integers(0, 2**32 - 1), tag=st.booleans(), sevenf=st.integers(0, 3))
def test_matches_oracle_on_random_transfers(built_lib, seed, n_blocks, tail, offset, tag, sevenf):
    rng = np.random.default_rng(seed)
    t = make_transfer(rng, n_blocks, rng.integers(1, 2**63, 3).tolist(), tag=tag)
    t = np.concatenate([t, rng.integers(0, 256, 2 * tail, dtype=np.uint8)])   # ragged last block
    if sevenf & 1:
        t[10] = 0x7F
    if sevenf & 2:
        t[11] = 0x7F
    both(t, offset)

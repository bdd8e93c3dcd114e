"""The one part of the path that can be pinned against the REFERENCE ITSELF in this image: the frequency table.

/root/reference/frequencyTable.cpp needs only the standard library, so `make -C oracle ref` compiles it from where it
lies (with oracle/ref_binding.cpp) into oracle/_ref/.  tests/golden/frequency_table_ref.npz was generated from that
build (tests/golden/make_frequency_table_ref.py) and is what these tests always check; where oracle/_ref is present
the live reference is compared on random sweeps too.  All of it is CPU work (`-m "not gpu"`): the built file is loaded
lazily, by those tests only, so a `-m gpu` run on the GPU box never maps it.
Checked against it: the oracle's restatement, the product's scn_frequency_table (GPU-free entry point of the C-ABI
library) and the host mirror's FrequencyTable class (tables and the GetCurrent/GetNext/GetIsScanStart walk)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from oracle import oracle
from scanner_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = np.load(os.path.join(ROOT, "tests", "golden", "frequency_table_ref.npz"))
CASES = [tuple(c) for c in GOLD["cases"]]


@pytest.mark.parametrize("k", range(len(CASES)))
def test_oracle_and_product_match_reference_fixture(built_lib, k):
    fs, start, stop, bw, dc = CASES[k]
    want = GOLD[f"table_{k}"]
    got_oracle = oracle.frequency_table(int(fs), start, stop, bw, dc)
    first, got_product = capi.frequency_table(int(fs), start, stop, bw, dc)
    assert first == 0
    assert got_oracle.tobytes() == want.tobytes()       # bit-exact doubles
    assert got_product.tobytes() == want.tobytes()
    # shards of the product's table are contiguous pieces of the same table
    if len(want) >= 8:
        pieces = [capi.frequency_table(int(fs), start, stop, bw, dc, shard=s, n_shards=8) for s in range(8)]
        assert [p[0] for p in pieces] == list(np.cumsum([0] + [len(p[1]) for p in pieces[:-1]]))
        assert np.concatenate([p[1] for p in pieces]).tobytes() == want.tobytes()


@pytest.fixture(scope="module")
def host_table_lib(built_lib, tmp_path_factory):
    """The host mirror's FrequencyTable behind the same C binding the reference build uses."""
    out = tmp_path_factory.mktemp("hostft") / "libhost_frequency_table.so"
    host = os.path.join(ROOT, "scanner_amd", "host")
    subprocess.check_call(["g++", "-std=gnu++11", "-O2", "-fPIC", "-shared", "-I", host, "-o", str(out),
                           os.path.join(ROOT, "oracle", "ref_binding.cpp"), os.path.join(host, "frequencyTable.cpp"),
                           "-L" + os.path.join(ROOT, "scanner_amd"), "-lscanner_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "scanner_amd")])
    L = C.CDLL(str(out))
    u32, dbl, vp = C.c_uint32, C.c_double, C.c_void_p
    L.ref_frequency_table.restype = u32
    L.ref_frequency_table.argtypes = [u32, dbl, dbl, dbl, dbl, vp, u32]
    L.ref_frequency_walk.argtypes = [u32, dbl, dbl, dbl, dbl, u32, vp, vp, vp]
    return L


def _walk(L, fs, start, stop, bw, dc, steps):
    f, it, ss = np.empty(steps), np.empty(steps, np.uint32), np.empty(steps, np.uint8)
    L.ref_frequency_walk(int(fs), start, stop, bw, dc, steps, f.ctypes.data_as(C.c_void_p), it.ctypes.data_as(C.c_void_p),
                         ss.ctypes.data_as(C.c_void_p))
    return f, it, ss


@pytest.mark.parametrize("k", range(len(CASES)))
def test_host_mirror_walks_like_the_reference(host_table_lib, k):
    fs, start, stop, bw, dc = CASES[k]
    want = GOLD[f"table_{k}"]
    n = host_table_lib.ref_frequency_table(int(fs), start, stop, bw, dc, None, 0)
    assert n == len(want)
    if not n:
        return
    got = np.empty(n)
    host_table_lib.ref_frequency_table(int(fs), start, stop, bw, dc, got.ctypes.data_as(C.c_void_p), n)
    assert got.tobytes() == want.tobytes()
    wf, wit, wss = GOLD[f"walk_f_{k}"], GOLD[f"walk_it_{k}"], GOLD[f"walk_ss_{k}"]
    f, it, ss = _walk(host_table_lib, fs, start, stop, bw, dc, len(wf))
    assert f.tobytes() == wf.tobytes() and np.array_equal(it, wit) and np.array_equal(ss, wss)


needs_ref = pytest.mark.skipif(not oracle.ref_available(), reason="oracle/_ref not built (needs /root/reference)")


@needs_ref
@settings(max_examples=150, deadline=None)
@given(fs=st.sampled_from([2400000, 8000000, 10000000, 12500000, 20000000, 61440000]),
       start=st.floats(1e6, 5.9e9), span=st.floats(1e5, 4e8), bw=st.sampled_from([0.5, 0.75, 0.8, 1.0]),
       dc=st.sampled_from([0.0, 0.0, 0.1, 0.25]))
def test_live_reference_on_random_sweeps(built_lib, host_table_lib, fs, start, span, bw, dc):
    stop = start + span
    step = (bw - dc) / 2 if dc > 0 else bw
    f1 = start + bw / 2 * fs
    count = 0
    while f1 + count * step * float(fs) < stop:
        count += 1
    # the reference asserts count == ceil((stop - f1)/(step*fs)) (frequencyTable.cpp:29); stay where that holds
    if count != int(np.ceil((stop - f1) / (step * fs))) or count > 4000:
        return
    want = oracle.ref_frequency_table(fs, start, stop, bw, dc)
    assert oracle.frequency_table(fs, start, stop, bw, dc).tobytes() == want.tobytes()
    assert capi.frequency_table(fs, start, stop, bw, dc)[1].tobytes() == want.tobytes()
    if len(want):
        steps = min(2 * len(want) + 3, 200)
        wf, wit, wss = oracle.ref_frequency_walk(fs, start, stop, bw, dc, steps)
        f, it, ss = _walk(host_table_lib, fs, start, stop, bw, dc, steps)
        assert f.tobytes() == wf.tobytes() and np.array_equal(it, wit) and np.array_equal(ss, wss)

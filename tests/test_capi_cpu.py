"""CPU-side checks of the C-ABI library: it builds, loads, exports every symbol the header
declares, and its GPU-free entry points behave.  No compute calls (no GPU here)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from scanner_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "scanner_hip.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(scn_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(built_lib):
    names = declared_symbols()
    assert len(names) >= 14
    out = subprocess.check_output(["nm", "-D", "--defined-only", built_lib], text=True)
    exported = set(re.findall(r" T (scn_\w+)", out))
    missing = [n for n in names if n not in exported]
    assert not missing, f"declared in scanner_hip.h but not exported: {missing}"
    # ... and the ctypes twin binds exactly that set
    assert sorted(capi.SYMBOLS) == names


def test_library_exports_nothing_but_the_header(built_lib):
    """A maintainer links this library into a C++ program: its dynamic symbol table is the C-ABI and nothing else -- no kernel
    launch stubs, no launchers, no cross-file helpers, no libstdc++ template instantiations (hidden visibility + SCN_API + a
    version script generated from the header, scanner_amd/build.py)."""
    out = subprocess.check_output(["nm", "-D", "--defined-only", built_lib], text=True)
    defined = sorted(ln.split()[-1] for ln in out.splitlines() if ln.strip())
    assert defined == declared_symbols(), sorted(set(defined) ^ set(declared_symbols()))
    src = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    for name in declared_symbols():     # every declaration carries the export macro
        assert re.search(r"\bSCN_API\b[^;{]*?\b%s\s*\(" % name, src), name


def test_python_constants_follow_the_header():
    """capi.py restates a few of the header's constants for ctypes callers: they must be the header's (SCN_NUM_SLOTS went from
    2 to 4 in round 3; a stale copy would make Plan refuse -- or worse, mis-size -- the slots the library has)."""
    src = open(HEADER).read()

    def macro(name):
        return int(re.search(r"#define\s+%s\s+(\d+)" % name, src).group(1))

    assert capi.NUM_SLOTS == macro("SCN_NUM_SLOTS")
    assert capi.ABI_VERSION == macro("SCN_ABI_VERSION")
    enum = dict((k, int(v)) for k, v in re.findall(r"(SCN_(?:OUT_SPECTRUM|OUT_HITS|PLAN_OVERLAP_SLOTS))\s*=\s*(\d+)u", src))
    assert (capi.OUT_SPECTRUM, capi.OUT_HITS, capi.PLAN_OVERLAP_SLOTS) == (
        enum["SCN_OUT_SPECTRUM"], enum["SCN_OUT_HITS"], enum["SCN_PLAN_OVERLAP_SLOTS"])


def test_header_compiles_as_c_and_cxx(tmp_path):
    for comp, std, ext in (("gcc", "-std=c99", "c"), ("g++", "-std=c++11", "cpp")):
        f = tmp_path / f"t.{ext}"
        f.write_text('#include "scanner_hip.h"\nint main(void){ scn_plan_desc d; d.struct_size = sizeof d; '
                     'return (int)(sizeof(scn_hit) != 24) + (int)(d.struct_size != %d); }\n'
                     % C.sizeof(capi.PlanDesc))
        exe = tmp_path / f"t_{ext}"
        subprocess.check_call([comp, std, "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(f), "-o",
                               str(exe)])
        assert subprocess.call([str(exe)]) == 0, "struct layout differs between header and ctypes binding"


def test_loads_and_reports_errors(built_lib):
    L = capi.lib()
    assert L.scn_abi_version() == capi.ABI_VERSION
    assert L.scn_error_name(0) == b"SCN_OK" and L.scn_error_name(capi.E_TRUNCATED) == b"SCN_E_TRUNCATED"
    assert L.scn_error_name(1234) == b"SCN_E_UNKNOWN"
    h = C.c_void_p()
    d = capi.PlanDesc()
    assert L.scn_plan_create(C.byref(d), C.byref(h)) == capi.E_INVALID     # struct_size not set
    assert b"struct_size" in L.scn_last_error()
    assert L.scn_plan_create(None, C.byref(h)) == capi.E_INVALID
    assert L.scn_plan_destroy(None) == capi.OK
    d.struct_size = C.sizeof(capi.PlanDesc)
    d.sample_rate, d.sample_kind, d.max_batch = 8000000, capi.KIND_FLOAT_COMPLEX, 4
    for bad in (8, 15, 65537, 131072):                                     # too small, too large
        d.n = bad
        assert L.scn_plan_create(C.byref(d), C.byref(h)) == capi.E_INVALID and b"16 to 65536" in L.scn_last_error()
    d.n, d.sample_kind = 4096, 9
    assert L.scn_plan_create(C.byref(d), C.byref(h)) == capi.E_INVALID
    # the frequency-table entry points refuse a null plan before they touch the device
    assert L.scn_plan_set_table(None, None, 0) == capi.E_INVALID and b"null plan" in L.scn_last_error()
    assert L.scn_submit_indexed(None, 0, 1, 0, None) == capi.E_INVALID
    assert L.scn_submit_device_indexed(None, 0, None, 1, 0, None, None) == capi.E_INVALID


def test_size_paths(built_lib):
    """scn_size_path needs no device: every power of two from 16 to 16384 has a fused kernel, and so have the 5-smooth sizes of
    scn_mixed_plans.h (1000 ... 16000); 32768 / 65536 the four-step pair, the other sizes from 17 to 65535 Bluestein, nothing
    else is planned (the GPU suite walks the fused list)."""
    for k in range(4, 15):
        assert capi.size_path(1 << k) == capi.PATH_FUSED, 1 << k
    for n in (1000, 1536, 3000, 5000, 6000, 10000, 10240, 12000, 15360, 16000):
        assert capi.size_path(n) == capi.PATH_FUSED, n
    assert capi.size_path(32768) == capi.size_path(65536) == capi.PATH_FOUR_STEP
    for n in (17, 100, 1001, 1023, 4097, 11000, 65535):
        assert capi.size_path(n) == capi.PATH_BLUESTEIN, n
    for n in (0, 1, 8, 15, 65537, 1 << 17):
        assert capi.size_path(n) == capi.PATH_UNSUPPORTED, n


def test_no_gpu_means_loud_failure(built_lib):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from scanner_amd import Plan

    with pytest.raises(capi.ScannerError) as e:
        Plan(4096, max_batch=1)
    assert e.value.status in (capi.E_NO_DEVICE, capi.E_HIP)


def test_frequency_table_and_shards(built_lib, oracle_mod):
    ref = oracle_mod.frequency_table(8000000, 0.0, 16384 * 6e6)
    first, whole = capi.frequency_table(8000000, 0.0, 16384 * 6e6)
    assert first == 0 and np.array_equal(whole, ref)
    parts = [capi.frequency_table(8000000, 0.0, 16384 * 6e6, shard=r, n_shards=8) for r in range(8)]
    assert [p[0] for p in parts] == [2048 * r for r in range(8)]
    assert np.array_equal(np.concatenate([p[1] for p in parts]), ref)     # contiguous, rank-major
    # ragged: 10 centres over 4 shards
    ref2 = oracle_mod.frequency_table(8000000, 88e6, 88e6 + 10 * 6e6)
    parts = [capi.frequency_table(8000000, 88e6, 88e6 + 10 * 6e6, shard=r, n_shards=4) for r in range(4)]
    assert sum(len(p[1]) for p in parts) == len(ref2) == 10
    assert np.array_equal(np.concatenate([p[1] for p in parts]), ref2)
    assert len(capi.frequency_table(8000000, 88e6, 0.0)[1]) == 1
    with pytest.raises(capi.ScannerError):
        capi.frequency_table(8000000, 0.0, 1e9, shard=2, n_shards=2)


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under scanner_amd/ may reference it."""
    pkg = os.path.join(ROOT, "scanner_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "scn_oracle" not in text and "import oracle" not in text and "from oracle" not in text, f

"""Builds scanner_amd/libscanner_hip.so (HIP kernels + C-ABI) for gfx950 with hipcc.

In-tree on purpose: the .so is git-ignored but travels to the GPU box with the repo
snapshot.  hipcc cross-compiles without a GPU, so this also runs on CPU-only hosts.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libscanner_hip.so")
SOURCES = ["scn_kernels.hip", "scn_mixed.hip", "scn_generic.hip", "scn_big.hip", "scn_hits.hip", "scn_welch.hip", "scn_gather.hip", "scn_api.hip"]
HEADERS = ["scn_kernels.h", "scn_device.h", "scn_mixed_dft.h", "scn_mixed_plans.h", "scn_gather_protocol.h", os.path.join("..", "..", "include", "scanner_hip.h")]
ARCH = "gfx950"
# The library's dynamic symbol table is the C-ABI and nothing else: everything is compiled with hidden visibility, the entry
# points of include/scanner_hip.h carry SCN_API (default visibility while SCN_BUILDING_LIBRARY is defined), and the link takes
# a version script generated from the header's own declarations (which also keeps libstdc++'s weak template instantiations --
# default visibility whatever the flags say -- out of the table).  tests/test_capi_cpu.py holds `nm -D` to the header.
EXTRA_FLAGS = ["-fvisibility=hidden", "-fvisibility-inlines-hidden", "-DSCN_BUILDING_LIBRARY"]
PUBLIC_HEADER = os.path.join(HERE, "..", "include", "scanner_hip.h")
MAP_FILE = os.path.join(HERE, "libscanner_hip.map")
LINK_FLAGS = ["-Wl,--version-script=" + MAP_FILE]


def declared_symbols():
    """The functions include/scanner_hip.h declares (every one carries SCN_API)."""
    import re

    with open(PUBLIC_HEADER) as fh:
        text = re.sub(r"/\*.*?\*/", "", fh.read(), flags=re.S)
    return sorted(set(re.findall(r"\bSCN_API\b[^;{]*?\b(scn_\w+)\s*\(", text)))


def write_map():
    with open(MAP_FILE, "w") as fh:
        fh.write("{\n  global:\n" + "".join(f"    {n};\n" for n in declared_symbols()) + "  local:\n    *;\n};\n")
# files compiled once per value of a macro, side by side, each translation unit instantiating one group of sizes
SPLIT = {"scn_kernels.hip": ("SCN_TU", 8), "scn_mixed.hip": ("SCN_MIXED_TU", 8)}


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; the HIP path cannot be built")
    return exe


def source_hash():
    """sha256 (first 16 hex digits) over the HIP sources and headers the library is built from: names the BUILD a profile
    was taken on (profiles/measured_shapes.json) so that bench.py never prints another build's traffic as this one's."""
    import hashlib

    h = hashlib.sha256()
    for f in sorted(SOURCES + HEADERS):
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    with open(os.path.abspath(__file__), "rb") as fh:  # the flags, ARCH and the split of the translation units live in this file
        h.update(b"build.py\0" + fh.read())
    return h.hexdigest()[:16]


HASH_FILE = LIB + ".hash"  # source_hash() of the sources the library beside it was built from


def _stale():
    """The library is current iff the hash recorded beside it at build time equals the hash of the sources as they are now --
    the same identity the profiling evidence is stamped with (profiles/measured_shapes.json).  Modification times are not
    consulted: a pushed tree whose .so is newer than an edited source would otherwise run stale kernels under a current hash."""
    if not os.path.exists(LIB) or not os.path.exists(HASH_FILE):
        return True
    with open(HASH_FILE) as fh:
        return fh.read().strip() != source_hash()


def build(force=False, verbose=False, defines=(), out=None):
    """Compile every HIP source for gfx950 into one shared library; returns its path.
    `defines` / `out` build an experiment variant (scripts/build_variants.py) next to the product library."""
    target = out or LIB
    if not out and not force and not _stale():
        return LIB
    import fcntl

    with open(LIB + ".lock", "w") as lock:  # several ranks (or pytest workers) finding the library stale build it ONCE, one after the other
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not out and not force and not _stale():
            return LIB
        return _build_locked(target, verbose, defines, out)


def _build_locked(target, verbose, defines, out):
    built_from = source_hash()
    write_map()
    objs, cmds = [], []
    tag = "" if not out else "." + os.path.basename(out).replace(".so", "")
    units = [(src, None) for src in SOURCES if src not in SPLIT] + [(src, tu) for src, (_, count) in SPLIT.items() for tu in range(count)]
    for src, tu in sorted(units, key=lambda u: u[1] is None):  # the fused-kernel units first: they take longest
        suffix = tag + ("" if tu is None else f".tu{tu}") + ".o"
        obj = os.path.join(os.path.dirname(out) if out else CSRC, src.replace(".hip", suffix))  # variants keep their objects beside them
        # -fno-slp-vectorize: keep the FFT butterflies as scalar f32 ops.  On gfx950 a packed
        # v_pk_*_f32 issues in the same 4 cycles as two scalar ops, and the SLP-packed stream
        # needs ~180 extra v_mov/v_pk_mov per FFT to pair registers (measured: 801 vs 668 VALU
        # instructions in the loop body).
        cmd = [hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-unused-const-variable",
               "-fno-slp-vectorize", *EXTRA_FLAGS, *[f"-D{d}" for d in defines], *([] if tu is None else [f"-D{SPLIT[src][0]}={tu}"]),
               "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        cmds.append(cmd)
        objs.append(obj)
    from concurrent.futures import ThreadPoolExecutor

    with ThreadPoolExecutor(max_workers=min(len(cmds), os.cpu_count() or 1)) as pool:  # one hipcc per translation unit, side by side
        list(pool.map(subprocess.check_call, cmds))
    cmd = [hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", *LINK_FLAGS, "-o", target] + objs + ["-ldl"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    if not out:
        with open(HASH_FILE, "w") as fh:
            fh.write(built_from + "\n")
    return target


HOST = os.path.join(HERE, "host")
HOST_LIB = os.path.join(HERE, "libscanner_host.so")
HOST_DEMO = os.path.join(HOST, "scan_synth")
ABI_BENCH = os.path.join(HOST, "abi_bench")
HOST_SOURCES = ["frequencyTable.cpp", "messageQueue.cpp", "signalSource.cpp", "syntheticSource.cpp", "fileSource.cpp",
                "processInterface.cpp", "sampleBuffer.cpp", "process.cpp"]


def build_host(force=False, verbose=False):
    """The C++ host library (kept SignalSource / SampleQueue / ProcessSamples surface on top of the
    C-ABI) and the scan_synth driver.  Plain g++ -std=gnu++11 like the reference (Makefile:23)."""
    build(force=False)
    srcs = [os.path.join(HOST, f) for f in HOST_SOURCES]
    deps = srcs + [os.path.join(HOST, f) for f in os.listdir(HOST) if f.endswith(".h")] + [LIB]
    if force or not os.path.exists(HOST_LIB) or any(os.path.getmtime(d) > os.path.getmtime(HOST_LIB) for d in deps):
        cmd = ["g++", "-std=gnu++11", "-O2", "-g", "-fPIC", "-shared", "-Wall", "-pthread", "-o", HOST_LIB] + srcs + \
              ["-L" + HERE, "-lscanner_hip", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    demo_src = os.path.join(HOST, "scan_synth.cpp")
    if force or not os.path.exists(HOST_DEMO) or \
            max(os.path.getmtime(demo_src), os.path.getmtime(HOST_LIB)) > os.path.getmtime(HOST_DEMO):
        cmd = ["g++", "-std=gnu++11", "-O2", "-g", "-Wall", "-pthread", "-o", HOST_DEMO, demo_src, "-L" + HERE,
               "-lscanner_host", "-lscanner_hip", "-Wl,-rpath,$ORIGIN/.."]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    # abi_bench: the C-ABI's step loop from a C++ process (the system HIP runtime, not the one torch bundles)
    bench_src = os.path.join(HOST, "abi_bench.cpp")
    if force or not os.path.exists(ABI_BENCH) or os.path.getmtime(bench_src) > os.path.getmtime(ABI_BENCH) or \
            os.path.getmtime(os.path.join(HERE, "..", "include", "scanner_hip.h")) > os.path.getmtime(ABI_BENCH):
        rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
        cmd = ["g++", "-std=gnu++11", "-O2", "-g", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(rocm, "include"), "-o", ABI_BENCH,
               bench_src, "-L" + os.path.join(rocm, "lib"), "-lamdhip64", "-ldl", "-Wl,-rpath," + os.path.join(rocm, "lib")]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return HOST_LIB, HOST_DEMO


if __name__ == "__main__":
    build_host(force="--force" in sys.argv, verbose=True)
    print(build(force="--force" in sys.argv, verbose=True))

"""The mixed-radix fused kernels (scanner_amd/csrc/scn_mixed.hip) as far as a CPU can hold them: the in-register DFTs of every
5-smooth length against the DFT sum (the header is plain arithmetic and compiles for the host), the three-pass index algebra of
every supported size emulated in numpy against numpy.fft, the generated plan header in step with its generator, and the oracle's
factored DFT (what the GPU tests of these sizes are judged against) identical to the DFT sum as written."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_in_register_dfts_against_the_dft_sum(tmp_path):
    exe = tmp_path / "test_mixed_dft"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", str(exe), os.path.join(ROOT, "tests", "cpp", "test_mixed_dft.cpp")])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "mixed dft tests ok" in out.stdout, out.stdout[-2000:]
    for r in (8, 10, 12, 15, 16, 18, 20, 24, 25, 32):   # every radix scn_mixed_plans.h uses
        assert f"dft<{r:2d}>" in out.stdout
    for r in (16, 20, 24):                            # ... and the last-pass radices of the sizes beyond 10000, in double
        assert f"dft<{r:2d}> in double" in out.stdout


def test_plans_verify_and_the_header_is_the_generators_output(tmp_path):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "mixed_plan.py"), "--all"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]          # (asserts inside: every size's three passes == numpy.fft to 1e-12, pitch costs as recorded)
    rows = [l for l in out.stdout.splitlines() if " = " in l and "threads" in l]
    header = open(os.path.join(ROOT, "scanner_amd", "csrc", "scn_mixed_plans.h")).read()
    assert len(rows) == header.count("\n  X(") == 34
    for l in rows:
        n, r1, r2, r3 = (int(v) for v in l.replace("=", " ").replace("x", " ").split()[:4])
        assert f"  X({n}, {r1}, {r2}, {r3}, " in header and r1 * r2 * r3 == n and "emulated vs numpy.fft" in l
        if n <= 10000:
            assert r1 <= min(r2, r3) and r2 * r3 <= 512      # one virtual thread per thread, the smallest radix first
        else:
            assert r3 <= min(r1, r2) and "x 2 virtual" in l  # SCN_MIXED_BIG_PLANS: the smallest radix last (pass 3 in double)


def test_oracle_factored_dft_is_the_dft_sum(oracle_mod):
    """lengths that are not powers of two: the oracle factors the DFT sum over the prime factors of n (fast enough for 5000-buffer
    GPU tests); the same sum evaluated as written, O(n^2), gives bit-identical float spectra"""
    from scanner_amd import synth

    O = oracle_mod
    for n in (17, 30, 1000, 1023, 3000, 4097, 6000):
        x = synth.cfloat_batch(n, 3, seed=n, sigma=0.1)
        with O.direct_dft():
            a = O.Oracle(n, 8000000, 1e9).run(x)[0]
        b = O.Oracle(n, 8000000, 1e9).run(x)[0]
        assert np.array_equal(a, b), n

"""The register budget of the shipped kernels, held on CPU (hipcc cross-compiles; the figures come from the code objects' own
metadata): a kernel that starts to spill shows up HERE, not as an unexplained slowdown on the GPU.  Round 4 shipped twelve
spilling specialisations nobody had looked at: the planar-int16 16 .. 128-point kernels (two 16-bit loads
packed in registers held 32 prefetch registers instead of 16: one 32-bit load now) and the hits-only 16384-point kernels
(a select over gmax[4] that hipcc had turned into an indexed read of a private array: 32 bytes of scratch)."""
import importlib.util
import os

from scanner_amd import build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# kernel -> (VGPRs spilled, scratch bytes) that are accepted, each with its reason
ALLOWED = {
    # five window / twiddle values spilled ONCE in the prologue and reloaded once before the buffer loop (ISA: every scratch
    # access sits in front of the loop's first barrier): nothing per buffer
    "void scn_fft8k_kernel<4, false, false, true>(ScnFftArgs)": (5, 24),
}


def test_no_kernel_spills_beyond_its_stated_budget():
    build.build()
    spec = importlib.util.spec_from_file_location("kernel_regs", os.path.join(ROOT, "scripts", "kernel_regs.py"))
    kr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kr)
    table = kr.table(build.LIB)
    assert len(table) > 300 and all(k["vgpr"] > 0 for k in table)          # the metadata was found and parsed
    assert any(k["name"].startswith("void scn_fft_kernel<16, 4, false, true, true>") for k in table)
    over = []
    for k in table:
        spill, scratch = ALLOWED.get(k["name"], (0, 0))
        if k["spill"] > spill or k["scratch"] > scratch:
            over.append((k["name"], k["spill"], k["scratch"]))
    assert not over, "kernels spilling beyond the budget (scripts/kernel_regs.py): %r" % over
    # the occupancy the launchers count on: the 4096-point kernels at three waves per SIMD (<= 168 VGPRs), 8192 / 16384 at two
    for k in table:
        if k["name"].startswith("void scn_fft_kernel<"):
            assert k["vgpr"] <= 168, k
        if k["name"].startswith(("void scn_fft8k_kernel<", "void scn_fft16k2_kernel<")):
            assert k["vgpr"] <= 256, k

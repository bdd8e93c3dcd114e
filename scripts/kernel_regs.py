"""VGPR / spill / LDS figures of every kernel in the built library (from the code objects' metadata notes):
   python scripts/kernel_regs.py [substring]      (SCN_LIB selects a variant build)
table(lib) returns the same as a list of dicts (tests/test_kernel_resources_cpu.py holds the spill budget against it)."""
import os, re, subprocess, sys, tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
DEFAULT_LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scanner_amd", "libscanner_hip.so")


def table(lib=None):
    lib = lib or os.environ.get("SCN_LIB") or DEFAULT_LIB
    with tempfile.TemporaryDirectory() as d:
        # .hip_fatbin holds one offload bundle per translation unit, back to back
        subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, f"{d}/fat.bin"])
        blob = open(f"{d}/fat.bin", "rb").read()
        magic = b"__CLANG_OFFLOAD_BUNDLE__"
        starts = [m.start() for m in re.finditer(magic, blob)] + [len(blob)]
        txt = ""
        for k in range(len(starts) - 1):
            open(f"{d}/b{k}.bin", "wb").write(blob[starts[k]:starts[k + 1]])
            r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                                f"--input={d}/b{k}.bin", f"--output={d}/k{k}.co"], stderr=subprocess.DEVNULL)
            if r.returncode == 0 and os.path.getsize(f"{d}/k{k}.co"):
                txt += subprocess.check_output([f"{LLVM}/llvm-readelf", "--notes", f"{d}/k{k}.co"], text=True)
    blocks = txt.split("  - .agpr_count:")[1:]
    names = subprocess.check_output(["c++filt"], input="\n".join((re.search(r"\.name:\s+(\S+)", b) or [None, "?"])[1] for b in blocks), text=True).splitlines()
    out = []
    for blk, name in zip(blocks, names):
        g = lambda k: int((re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "-1"])[1])  # noqa: E731
        out.append({"name": name.strip(), "vgpr": g("vgpr_count"), "spill": g("vgpr_spill_count"), "sgpr": g("sgpr_count"),
                    "sgpr_spill": g("sgpr_spill_count"), "scratch": g("private_segment_fixed_size"), "lds": g("group_segment_fixed_size")})
    return out


if __name__ == "__main__":
    want = sys.argv[1] if len(sys.argv) > 1 else ""
    for k in table():
        if want in k["name"]:
            print(f"vgpr {k['vgpr']:4d} spill {k['spill']:3d} sgpr {k['sgpr']:4d} scratch {k['scratch']:5d}  {k['name'][:150]}")

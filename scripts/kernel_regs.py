"""VGPR / spill / LDS figures of every kernel in the built library (from the code objects' metadata notes):
   python scripts/kernel_regs.py [substring]      (SCN_LIB selects a variant build)"""
import os, re, subprocess, sys, tempfile
lib = os.environ.get("SCN_LIB") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scanner_amd", "libscanner_hip.so")
want = sys.argv[1] if len(sys.argv) > 1 else ""
LLVM = "/opt/rocm/lib/llvm/bin"
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
for blk in txt.split("  - .agpr_count:")[1:]:
    g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
    name = subprocess.check_output(["c++filt", g("name")], text=True).strip()
    if want in name:
        print(f"vgpr {g('vgpr_count'):>4s} spill {g('vgpr_spill_count'):>3s} sgpr {g('sgpr_count'):>4s} scratch {g('private_segment_fixed_size'):>5s}  {name[:150]}")

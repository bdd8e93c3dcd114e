"""VGPR / spill / LDS figures of every kernel in the built library (from the code object's metadata notes):
   python scripts/kernel_regs.py [substring]"""
import re, subprocess, sys, os, tempfile
lib = os.environ.get("SCN_LIB") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scanner_amd", "libscanner_hip.so")
want = sys.argv[1] if len(sys.argv) > 1 else ""
with tempfile.TemporaryDirectory() as d:
    subprocess.check_call(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                           f"--input={lib}", f"--output={d}/k.co"], stderr=subprocess.DEVNULL) if False else None
    # the fat binary sits in .hip_fatbin of the shared library
    subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, f"{d}/fat.bin"])
    subprocess.check_call(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                           f"--input={d}/fat.bin", f"--output={d}/k.co"])
    txt = subprocess.check_output(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f"{d}/k.co"], text=True)
for blk in txt.split("  - .agpr_count:")[1:]:
    g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
    name = g("name")
    name = subprocess.check_output(["c++filt", name], text=True).strip()
    if want in name:
        print(f"vgpr {g('vgpr_count'):>4s} spill {g('vgpr_spill_count'):>3s} sgpr {g('sgpr_count'):>4s} lds {g('group_segment_fixed_size'):>6s} scratch {g('private_segment_fixed_size'):>5s}  {name[:150]}")

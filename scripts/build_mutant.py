"""Builds a MUTANT of the HIP library: one literal edit in one source file, everything else the product's own objects.
   python scripts/build_mutant.py <name> <file.hip> <old text> <new text>   -> scanner_amd/variants/lib_<name>.so
The edit must match exactly once.  Used to show that a test can see the fault it is there for (mutation check): run the test
with SCN_LIB=scanner_amd/variants/lib_<name>.so and expect it to FAIL.  The product source is never modified."""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scanner_amd import build  # noqa: E402

name, fname, old, new = sys.argv[1:5]
build.build()  # the product's objects (and library) are current
src = open(os.path.join(build.CSRC, fname)).read()
assert src.count(old) == 1, f"{old!r} occurs {src.count(old)} times in {fname}"
vdir = os.path.join(build.HERE, "variants")
os.makedirs(vdir, exist_ok=True)
tmp = os.path.join(build.CSRC, f"_mutant_{name}_{fname}")
obj = os.path.join(vdir, f"{name}_{fname}.o")
assert fname not in build.SPLIT, "split translation units are not supported here"
try:
    with open(tmp, "w") as fh:
        fh.write(src.replace(old, new))
    subprocess.check_call([build.hipcc(), f"--offload-arch={build.ARCH}", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", *build.EXTRA_FLAGS,
                           "-c", tmp, "-o", obj])
finally:
    if os.path.exists(tmp):
        os.remove(tmp)
objs = [obj]
for s in build.SOURCES:
    if s == fname:
        continue
    if s in build.SPLIT:
        objs += [os.path.join(build.CSRC, s.replace(".hip", f".tu{tu}.o")) for tu in range(build.SPLIT[s][1])]
    else:
        objs.append(os.path.join(build.CSRC, s.replace(".hip", ".o")))
out = os.path.join(vdir, f"lib_{name}.so")
subprocess.check_call([build.hipcc(), f"--offload-arch={build.ARCH}", "-shared", "-fPIC", *build.LINK_FLAGS, "-o", out] + objs + ["-ldl"])
print(out)

"""Builds experiment variants of the HIP library side by side:
   python scripts/build_variants.py name=DEF1,DEF2=val ...   -> scanner_amd/variants/lib_<name>.so
Run one with  SCN_LIB=scanner_amd/variants/lib_<name>.so python scripts/var_bench.py"""
import os, sys
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scanner_amd import build

vdir = os.path.join(build.HERE, "variants")
os.makedirs(vdir, exist_ok=True)
jobs = []
for spec in sys.argv[1:]:
    name, _, defs = spec.partition("=")
    jobs.append((name, [d for d in defs.split(",") if d]))
def one(j):
    name, defs = j
    return build.build(force=True, defines=defs, out=os.path.join(vdir, f"lib_{name}.so"))
with ThreadPoolExecutor(4) as ex:
    for p in ex.map(one, jobs):
        print(p)

#!/bin/bash
# Welch experiments: variant libraries (scripts/build_variants.py w*) x SCN_EXP_WELCH_PARTS, 32 PSDs per submit
cd "$GRAFT_REPO_ROOT"
for v in "$@"; do
  for parts in ${PARTS:-1}; do
    r=$(SCN_LIB=scanner_amd/variants/lib_$v.so SCN_EXP_WELCH_PARTS=$parts python3 bench.py --welch --welch-psd 32 --steps 300 --warmup 20 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "$v parts=$parts  $r"
  done
done

#!/bin/bash
# 8192 points: the wide kernel with neighbouring virtual threads per lane (product) against the first form (variant nopair)
cd "$GRAFT_REPO_ROOT"
for lib in "" scanner_amd/variants/lib_nopair.so; do
  for shape in "8192 int16 4096" "8192 cfloat 4096" "8192 int8 4096"; do
    echo -n "lib=${lib:-product} $shape: "; SCN_LIB=$lib python3 scripts/loop_only.py 1500 0 $shape | tail -1
  done
done

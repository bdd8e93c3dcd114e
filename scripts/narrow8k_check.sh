#!/bin/bash
# 8192 points: the product's wide kernel (256 threads x 32 points) against the 512-thread form, without and with the
# 16-register prefetch for the integer formats (variants narrow8k / narrowpf of scripts/build_variants.py)
cd "$GRAFT_REPO_ROOT"
for lib in "" scanner_amd/variants/lib_narrow8k.so scanner_amd/variants/lib_narrowpf.so; do
  for shape in "8192 int16 4096" "8192 int8 4096"; do
    echo -n "lib=${lib:-product} $shape: "; SCN_LIB=$lib python3 scripts/loop_only.py 1500 0 $shape | tail -1
  done
done

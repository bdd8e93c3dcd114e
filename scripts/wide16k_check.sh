#!/bin/bash
# 16384 points: the wide kernel (512 threads x 32 points, register prefetch; product) against scn_fft_kernel<64> (variant narrow16k)
cd "$GRAFT_REPO_ROOT"
for lib in "" scanner_amd/variants/lib_narrow16k.so; do
  for shape in "16384 cfloat 2048" "16384 int16 2048"; do
    echo -n "lib=${lib:-product} $shape: "; SCN_LIB=$lib python3 scripts/loop_only.py 1000 0 $shape 2>/dev/null | tail -1
  done
done

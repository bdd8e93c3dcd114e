#!/bin/bash
# 16384 points, float input: the wide form with a partial prefetch (PFN of 16 two-sample loads; product = 8) vs scn_fft_kernel<64>
cd "$GRAFT_REPO_ROOT"
for lib in "" scanner_amd/variants/lib_f16narrow.so scanner_amd/variants/lib_pfn8.so scanner_amd/variants/lib_pfn10.so scanner_amd/variants/lib_pfn14.so ""; do
  echo -n "lib=${lib:-product(pfn8)} 16384 cfloat 2048: "; SCN_LIB=$lib python3 scripts/loop_only.py 1000 0 16384 cfloat 2048 2>/dev/null | tail -1
done

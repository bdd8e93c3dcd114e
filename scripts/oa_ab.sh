#!/bin/bash
mkdir -p gpurun_out/oa
python3 -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "256 or 512" 2>&1 | grep -E "passed|failed"
for l in product nolns product; do
  echo "== $l"
  if [ $l = product ]; then unset SCN_LIB; else export SCN_LIB=scanner_amd/variants/lib_$l.so; fi
  python3 -u scripts/sweep_all.py 256 512 2>&1 | grep -v amdgpu.ids | grep " F "
done 2>&1 | tee gpurun_out/oa/small.txt

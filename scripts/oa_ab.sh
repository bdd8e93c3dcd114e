#!/bin/bash
mkdir -p gpurun_out/oa
python3 -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "16384" 2>&1 | tail -2
for l in product noscan product; do
  echo "== $l"
  if [ $l = product ]; then unset SCN_LIB; else export SCN_LIB=scanner_amd/variants/lib_$l.so; fi
  python3 -u scripts/sweep_all.py 16384 2>&1 | grep -v amdgpu.ids | grep "^16384"
done 2>&1 | tee gpurun_out/oa/mask16k.txt

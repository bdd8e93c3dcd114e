#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out/r03p; O=gpurun_out/r03p
for rep in 1 2; do
for v in "" v3 r02; do
  lib=""; [ -n "$v" ] && lib=scanner_amd/variants/lib_$v.so
  echo "== ${v:-new}"
  SCN_LIB=$lib python3 scripts/mode_loop.py 16384 int16 2048 3 400 | tail -1
  SCN_LIB=$lib python3 -u scripts/sweep_all.py 4096 8192 2>&1 | grep "cfloat\| int16   F\|int8   F"
done
done
timeout 1500 python3 -m pytest tests/test_parity_gpu.py tests/test_fuzz_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1; grep -E "passed|failed" $O/pytest.txt | tail -1

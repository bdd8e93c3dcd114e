#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out/r03n; O=gpurun_out/r03n
for a in "8192 3 int16" "4096 3 cfloat"; do SCN_LIB=scanner_amd/variants/lib_stamps.so python3 scripts/stamp_profile.py $a 2>&1 | grep -v amdgpu.ids | grep "cycles per\|hit recording\|barrier 1\|entered"; done
for rep in 1 2; do
for v in "" v3 r02; do
  lib=""; [ -n "$v" ] && lib=scanner_amd/variants/lib_$v.so
  echo "== ${v:-new}"
  SCN_LIB=$lib python3 scripts/mode_loop.py 4096 cfloat 8192 3 800 | tail -1
  SCN_LIB=$lib python3 scripts/mode_loop.py 8192 int16 4096 3 800 | tail -1
  SCN_LIB=$lib python3 scripts/mode_loop.py 16384 cfloat 2048 3 400 | tail -1
  SCN_LIB=$lib python3 scripts/mode_loop.py 16384 int16 2048 3 400 | tail -1
  SCN_LIB=$lib python3 scripts/mode_loop.py 16384 int16 2048 3 400 14.0 | tail -1
  SCN_LIB=$lib python3 scripts/mode_loop.py 4096 cfloat 8192 3 800 6.0 | tail -1
done
done
timeout 1200 python3 -m pytest tests/test_parity_gpu.py tests/test_fuzz_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1; grep -E "passed|failed" $O/pytest.txt | tail -1

#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out/r03l; O=gpurun_out/r03l
timeout 1200 python3 -m pytest tests/test_parity_gpu.py tests/test_fuzz_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1; grep -E "passed|failed" $O/pytest.txt | tail -1
for v in "" prev; do
  lib=""; [ -n "$v" ] && lib=scanner_amd/variants/lib_$v.so
  echo "== ${v:-new}"
  SCN_LIB=$lib python3 scripts/mode_loop.py 4096 cfloat 8192 3 600 | tail -1
  SCN_LIB=$lib python3 scripts/mode_loop.py 8192 int16 4096 3 600 | tail -1
  SCN_LIB=$lib python3 scripts/mode_loop.py 4096 int16 8192 3 600 | tail -1
  SCN_LIB=$lib python3 scripts/mode_loop.py 16384 cfloat 2048 3 300 | tail -1
  SCN_LIB=$lib python3 scripts/mode_loop.py 4096 cfloat 8192 2 600 | tail -1
  SCN_LIB=$lib python3 scripts/mode_loop.py 4096 cfloat 8192 3 600 6.0 | tail -1
done
for f in 3; do SCN_LIB=scanner_amd/variants/lib_stamps.so python3 scripts/stamp_profile.py 8192 $f int16 2>&1 | grep -v amdgpu.ids | head -14 | tee $O/stamps_c3_flags$f.txt; done
SCN_LIB=scanner_amd/variants/lib_stamps.so python3 scripts/stamp_profile.py 4096 3 cfloat 2>&1 | grep -v amdgpu.ids | head -14 | tee $O/stamps_c2.txt

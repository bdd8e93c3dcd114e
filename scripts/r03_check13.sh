#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out/r03o; O=gpurun_out/r03o
timeout 1500 python3 -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; grep -E "passed|failed" $O/pytest.txt | tail -1
for a in "8192 3 int16" "4096 3 cfloat"; do SCN_LIB=scanner_amd/variants/lib_stamps.so python3 scripts/stamp_profile.py $a 2>&1 | grep -v amdgpu.ids | grep "cycles per\|hit recording\|barrier 1\|entered"; done
for v in "" v3 r02; do
  lib=""; [ -n "$v" ] && lib=scanner_amd/variants/lib_$v.so
  echo "== ${v:-new}"
  SCN_LIB=$lib python3 scripts/mode_loop.py 16384 cfloat 2048 3 400 | tail -1
  SCN_LIB=$lib python3 scripts/mode_loop.py 16384 int16 2048 3 400 | tail -1
  SCN_LIB=$lib python3 scripts/mode_loop.py 16384 int16 2048 3 400 14.0 | tail -1
  SCN_LIB=$lib python3 scripts/mode_loop.py 16384 int16 2048 2 400 14.0 | tail -1
  SCN_LIB=$lib python3 -u scripts/sweep_all.py 4096 8192 2>&1 | grep "cfloat\| int16   F\|int8   F"
done
python3 scripts/acc16k.py | tail -3

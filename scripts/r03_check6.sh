#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03h; O=gpurun_out/r03h
for v in "" p3d; do
  lib=""; [ -n "$v" ] && lib=scanner_amd/variants/lib_$v.so
  echo "== ${v:-product}"; SCN_LIB=$lib python3 -u scripts/sweep_all.py 16384 2>&1 | grep -v amdgpu.ids | tee $O/sweep_${v:-product}.txt
  SCN_LIB=$lib python3 scripts/acc16k.py 2>&1 | tail -3 | tee $O/acc16k_${v:-product}.txt
  for k in cfloat int16; do SCN_LIB=$lib python3 scripts/mode_loop.py 16384 $k 2048 3 300 12.0 | tail -1; done
done
SCN_LIB=scanner_amd/variants/lib_p3d.so timeout 900 python3 -m pytest tests/test_parity_gpu.py tests/test_fuzz_gpu.py -x -q -m gpu -k "16384 or fuzz or sizes" > $O/pytest_p3d.txt 2>&1; grep -E "passed|failed" $O/pytest_p3d.txt | tail -2
python3 -u scripts/sweep_all.py 256 512 2>&1 | grep -v amdgpu.ids | tee $O/sweep_small.txt
timeout 600 python3 -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "generic or 512 or 256" > $O/pytest_small.txt 2>&1; grep -E "passed|failed" $O/pytest_small.txt | tail -2

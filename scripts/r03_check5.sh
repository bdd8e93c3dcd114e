#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03g; O=gpurun_out/r03g
timeout 900 python3 -m pytest tests/test_parity_gpu.py tests/test_fuzz_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1; grep -E "passed|failed" $O/pytest.txt | tail -2
for v in "" p3d old16k; do
  lib=""; [ -n "$v" ] && lib=scanner_amd/variants/lib_$v.so
  echo "== ${v:-product}"; SCN_LIB=$lib python3 -u scripts/sweep_all.py 16384 2>&1 | grep -v amdgpu.ids | tee $O/sweep_${v:-product}.txt
  SCN_LIB=$lib python3 scripts/acc16k.py 2>&1 | tail -3 | tee $O/acc16k_${v:-product}.txt
done

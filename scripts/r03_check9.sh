#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out/r03k; O=gpurun_out/r03k
for f in 3 1; do SCN_LIB=scanner_amd/variants/lib_stamps.so python3 scripts/stamp_profile.py 8192 $f int16 2>&1 | grep -v amdgpu.ids | tee $O/stamps_c3_flags$f.txt; done
SCN_LIB=scanner_amd/variants/lib_stamps.so python3 scripts/stamp_profile.py 4096 3 cfloat 2>&1 | grep -v amdgpu.ids | tee $O/stamps_c2.txt
for r in 0 1; do python3 scripts/loop_only.py 600 $r | tail -1; done
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/rec_tl -- python3 scripts/loop_only.py 80 1 > /dev/null 2> $O/rec_tl.log
python3 scripts/timeline.py $O/rec_tl 5 | tee $O/records_timeline.txt
rm -rf $O/rec_tl
timeout 600 python3 -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "65536" > $O/pytest_65536.txt 2>&1; grep -E "passed|failed" $O/pytest_65536.txt | tail -1
FUZZ_SIZES=65536 python3 scripts/fuzz_parity.py 420 77 2>&1 | grep -v amdgpu.ids | tee $O/fuzz_65536.txt
python3 -u scripts/sweep_all.py 65536 2>&1 | grep -v amdgpu.ids | head -3 | tee $O/sweep_65536.txt

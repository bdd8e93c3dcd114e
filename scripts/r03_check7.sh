#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03i; O=gpurun_out/r03i
timeout 1800 python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; grep -E "passed|failed|rror" $O/pytest_gpu.txt | tail -5
python3 -u scripts/sweep_all.py 65536 2>&1 | grep -v amdgpu.ids | tee $O/sweep_65536.txt

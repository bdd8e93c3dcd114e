#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03j; O=gpurun_out/r03j
timeout 1800 python3 -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; grep -E "passed|failed|rror" $O/pytest_gpu.txt | tail -8
bash scripts/welch_chunk.sh 2>&1 | grep -v amdgpu.ids | tee $O/welch_chunk.txt

#!/bin/bash
# round 3 evidence in one call: every launch shape under rocprofv3 (+ PMC passes), the config table, the sweep, un-profiled bench lines
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
env | grep -i "sdma\|HSA_\|GPU_" > gpurun_out/env_r03.txt
SCN_PROF_MORE=1 bash scripts/prof_all.sh r03 > gpurun_out/prof_all_r03.txt 2>&1
bash scripts/other_configs.sh r03 > gpurun_out/other_r03.txt 2>&1
python3 -u scripts/sweep_all.py 256 512 1024 2048 4096 8192 16384 32768 65536 2>&1 | grep -v amdgpu.ids > gpurun_out/sweep_all_r03.txt
python3 bench.py > gpurun_out/bench_r03_final.json 2> gpurun_out/bench_r03_final.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_r03_driver.json 2>/dev/null
tail -2 gpurun_out/other_r03.txt; cat gpurun_out/bench_r03_final.json | head -c 1500

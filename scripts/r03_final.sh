#!/bin/bash
# the round's closing call on the final build: GPU suite, 30-minute fuzz, config table, the un-profiled bench lines
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out/r03_fuzz
python3 -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 > gpurun_out/pytest_gpu_r03.txt
python3 scripts/fuzz_parity.py 1800 61 > gpurun_out/r03_fuzz/fuzz_final.txt 2>&1
bash scripts/other_configs.sh r03 > gpurun_out/other_r03.txt 2>&1
python3 bench.py > gpurun_out/bench_r03_final.json 2> gpurun_out/bench_r03_final.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_r03_driver.json 2>/dev/null
cat gpurun_out/pytest_gpu_r03.txt; tail -2 gpurun_out/r03_fuzz/fuzz_final.txt; tail -3 gpurun_out/other_r03.txt; head -c 1200 gpurun_out/bench_r03_final.json

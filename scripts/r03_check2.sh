#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03c; O=gpurun_out/r03c
timeout 900 python3 -m pytest tests/test_parity_gpu.py -x -q -m gpu > $O/pytest_parity.txt 2>&1; tail -3 $O/pytest_parity.txt
python3 scripts/acc16k.py > $O/acc16k.txt 2>&1; tail -3 $O/acc16k.txt
python3 scripts/sweep_all.py 4096 16384 > $O/sweep.txt 2>&1; cat $O/sweep.txt

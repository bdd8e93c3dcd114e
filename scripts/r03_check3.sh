#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03e; O=gpurun_out/r03e
timeout 1500 python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
python3 -u scripts/sweep_all.py > $O/sweep_all.txt 2>&1; cat $O/sweep_all.txt
SCN_LIB=scanner_amd/variants/lib_r02.so python3 -u scripts/sweep_all.py 4096 8192 > $O/sweep_r02lib.txt 2>&1; cat $O/sweep_r02lib.txt
python3 scripts/acc16k.py > $O/acc16k.txt 2>&1; tail -3 $O/acc16k.txt

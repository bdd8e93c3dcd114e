#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03f; O=gpurun_out/r03f
python3 -u scripts/sweep_all.py 4096 8192 > $O/sweep_new.txt 2>&1; cat $O/sweep_new.txt
SCN_LIB=scanner_amd/variants/lib_r02.so python3 -u scripts/sweep_all.py 4096 8192 > $O/sweep_r02lib.txt 2>&1; cat $O/sweep_r02lib.txt
timeout 900 python3 -m pytest tests/test_parity_gpu.py tests/test_sweep_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt

#!/bin/bash
# round 3, first look: parity subset, 16384-point accuracy and timing with pass 3 in double (product) and in float (variant nop3d)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03b; O=gpurun_out/r03b
timeout 900 python3 -m pytest tests/test_parity_gpu.py -x -q -m gpu > $O/pytest_parity.txt 2>&1; tail -3 $O/pytest_parity.txt
python3 scripts/acc16k.py > $O/acc16k_p3d.txt 2>&1; tail -3 $O/acc16k_p3d.txt
SCN_LIB=scanner_amd/variants/lib_nop3d.so python3 scripts/acc16k.py > $O/acc16k_nop3d.txt 2>&1; tail -3 $O/acc16k_nop3d.txt
python3 scripts/sweep_all.py 4096 16384 > $O/sweep_p3d.txt 2>&1; cat $O/sweep_p3d.txt
SCN_LIB=scanner_amd/variants/lib_nop3d.so python3 scripts/sweep_all.py 16384 > $O/sweep_nop3d.txt 2>&1; cat $O/sweep_nop3d.txt

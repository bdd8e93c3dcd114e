#!/bin/bash
# round 5 evidence, one GPU call: full GPU suite, rocprofv3 + PMC for every launch shape, the bench lines, the sweep table, a fuzz slice
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
TAG=${1:-r05}
O=gpurun_out; mkdir -p $O
timeout 1800 python3 -m pytest tests -m gpu -q > $O/pytest_$TAG.txt 2>&1; echo "pytest rc $?"; tail -3 $O/pytest_$TAG.txt
SCN_PROF_MORE=1 timeout 4200 bash scripts/prof_all.sh $TAG > $O/prof_all_$TAG.txt 2>&1; echo "prof_all rc $?"
( time timeout 300 python3 bench.py > $O/bench_${TAG}_final.json 2> $O/bench_${TAG}_final.err ) 2> $O/bench_${TAG}_time.txt; echo "bench rc $?"; grep real $O/bench_${TAG}_time.txt
timeout 300 python3 bench.py --steps 20 --warmup 3 > $O/bench_${TAG}_steps20.json 2>/dev/null; echo "bench20 rc $?"
timeout 1500 bash scripts/other_configs.sh $TAG > $O/other_$TAG.txt 2>&1; echo "other rc $?"
timeout 900 python3 scripts/sweep_all.py 256 512 1024 2048 4096 8192 16384 > $O/sweep_all_$TAG.txt 2>&1; echo "sweep rc $?"
timeout 700 python3 scripts/fuzz_parity.py 600 51 > $O/fuzz_$TAG.txt 2>&1; echo "fuzz rc $?"; tail -5 $O/fuzz_$TAG.txt
bash scripts/host_path_rate.sh > $O/host_path_rate_$TAG.txt 2>&1; echo "host path rc $?"

#!/bin/bash
# round 5, first GPU call: the GPU suite on the round's groundwork, the default bench line (with `configs`), the sweep table
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r05a; mkdir -p $O
timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt
tail -5 $O/pytest.txt
timeout 300 python3 bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
timeout 300 python3 bench.py --steps 20 --warmup 3 > $O/bench20.json 2> $O/bench20.err; echo "bench20 rc $?"
timeout 900 python3 scripts/sweep_all.py 128 1024 2048 4096 8192 16384 > $O/sweep_all.txt 2>&1; echo "sweep rc $?"
cd scanner_amd/host && for m in "--hits-only 1 --depth 3" "--hits-only 1 --depth 4" "--hits-only 0 --depth 3"; do ./abi_bench --mode view $m --steps 300 2>/dev/null | tail -1; done > ../../$O/abi.txt

#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r05g; mkdir -p $O
SCN_LIB=$PWD/scanner_amd/variants/lib_big_nt.so timeout 600 python3 -m pytest tests/test_parity_gpu.py -x -q -k "generic_sizes or round3_kernels or dc_quirk_negative_mean and 32768" > $O/pytest.txt 2>&1; echo "parity rc $?"; tail -2 $O/pytest.txt
for rnd in 1 2; do
for v in default big_nt; do
  lib=$PWD/scanner_amd/variants/lib_$v.so; [ $v = default ] && lib=$PWD/scanner_amd/libscanner_hip.so
  for cfg in "65536 512 cfloat" "65536 512 int16" "32768 1024 cfloat"; do set -- $cfg
    SCN_LIB=$lib timeout 300 python3 bench.py --n $1 --batch $2 --kind $3 --no-cpu-baseline --no-records-leg --no-overlap-leg --no-copy-ref --no-hits-only-leg --steps 200 --warmup 20 2>/dev/null | tail -1 | \
      python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v $1 $3 round $rnd: %.1f us per step, %.1f Gs/s' % (d['ms_per_step']*1e3, d['value']/1e3))"
  done
done
done 2>&1 | tee $O/ab.txt

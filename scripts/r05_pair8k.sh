#!/bin/bash
# round 5: 8192 points as a pair of 4096-point transforms against the shipped wide kernel, one box
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r05d; mkdir -p $O
for v in pair8k pair8k2; do
  SCN_LIB=$PWD/scanner_amd/variants/lib_$v.so timeout 600 python3 -m pytest tests/test_dispatch_gpu.py -x -q -k "8192-int" > $O/pytest_$v.txt 2>&1; echo "$v parity rc $?"; tail -2 $O/pytest_$v.txt
done
for rnd in 1 2; do
for v in default pair8k pair8k2; do
  lib=$PWD/scanner_amd/variants/lib_$v.so; [ $v = default ] && lib=$PWD/scanner_amd/libscanner_hip.so
  for kind in int16 int8; do
    SCN_LIB=$lib timeout 300 python3 bench.py --n 8192 --kind $kind --batch 4096 --no-cpu-baseline --no-records-leg --no-overlap-leg --no-copy-ref --steps 500 --warmup 20 2>/dev/null | tail -1 | \
      python3 -c "import json,sys; d=json.loads(sys.stdin.read()); h=d['hits_only']; print('$v $kind round $rnd: spectrum+hits %.2f us per launch (events), %.1f Gs/s; hits-only %.2f us, %.1f Gs/s' % (d['roofline']['kernel_avg_ms']*1e3, d['value']/1e3, h['kernel_avg_ms']*1e3, h['value']/1e3))"
  done
done
done 2>&1 | tee $O/ab.txt
for v in default pair8k; do
  lib=$PWD/scanner_amd/variants/lib_$v.so; [ $v = default ] && lib=$PWD/scanner_amd/libscanner_hip.so
  SCN_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$v -- python3 bench.py --n 8192 --kind int16 --batch 4096 --no-cpu-baseline --no-records-leg --no-overlap-leg --no-copy-ref --steps 500 --warmup 20 > /dev/null 2>&1
  grep -h "scn_fft8k" $O/trace_$v/*/*kernel_stats.csv | head -3 | tee -a $O/ab.txt
  rm -rf $O/trace_$v
done

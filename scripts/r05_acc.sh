#!/bin/bash
# round 5: the accuracy tail of the mixed-radix kernels at the large sizes (strong tones), then a fuzz on exactly those sizes
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r05j; mkdir -p $O
timeout 900 python3 -m pytest tests/test_dispatch_gpu.py -x -q -k "12000 or 12288 or 14400 or 15000 or 16000" > $O/pytest_dispatch.txt 2>&1; echo "dispatch rc $?"; tail -2 $O/pytest_dispatch.txt
timeout 900 python3 scripts/acc16k.py 12000 12288 14400 15000 16000 2>&1 | grep -v amdgpu.ids | tee $O/acc.txt
for cfg in "12000 2796" "12288 2730" "14400 2330" "15000 2236" "16000 2097"; do set -- $cfg
  timeout 300 python3 bench.py --n $1 --batch $2 --no-cpu-baseline --no-records-leg --no-overlap-leg --no-copy-ref --steps 300 --warmup 20 2>/dev/null | tail -1 | \
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); h=d['hits_only']; print('$1 cfloat: %.1f us per step, %.1f Gs/s, frac %.4f; hits-only %.1f Gs/s' % (d['ms_per_step']*1e3, d['value']/1e3, d['roofline']['frac'], h['value']/1e3))"
done | tee $O/bench.txt
FUZZ_SIZES=12000,12288,14400,15000,16000 timeout 800 python3 scripts/fuzz_parity.py 600 79 2>&1 | tail -6 | tee $O/fuzz_mixed.txt

#!/bin/bash
# round 5: the twelve further mixed-radix sizes (3 * 2^k, 5 * 2^k, 15 * 2^k): parity, accuracy tail of the large ones, time
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r05k; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_dispatch_gpu.py -x -q -k "1280 or 1536 or 1920 or 2560 or 3072 or 3840 or 5120 or 6144 or 7680 or 10240 or 12800 or 15360 or documented" > $O/pytest_dispatch.txt 2>&1; echo "dispatch rc $?"; tail -2 $O/pytest_dispatch.txt
timeout 900 python3 scripts/acc16k.py 7680 10240 12800 15360 2>&1 | grep -v amdgpu.ids | tee $O/acc.txt
for n in 1280 1536 1920 2560 3072 3840 5120 6144 7680 10240 12800 15360; do
  timeout 300 python3 bench.py --n $n --batch $((33554432 / n)) --no-cpu-baseline --no-records-leg --no-overlap-leg --no-copy-ref --steps 300 --warmup 20 2>/dev/null | tail -1 | \
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); h=d['hits_only']; print('$n cfloat: %.1f us per step, %.1f Gs/s, frac %.4f; hits-only %.1f Gs/s' % (d['ms_per_step']*1e3, d['value']/1e3, d['roofline']['frac'], h['value']/1e3))"
done | tee $O/bench.txt
FUZZ_SIZES=1280,1536,1920,2560,3072,3840,5120,6144,7680,10240,12800,15360 timeout 500 python3 scripts/fuzz_parity.py 300 83 2>&1 | tail -4 | tee $O/fuzz.txt

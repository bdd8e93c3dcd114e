#!/bin/bash
# round 5: does a smaller Welch submit (work buffer inside the 256 MiB Infinity Cache) pay now that the input stream is non-temporal?
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r05h; mkdir -p $O
for rnd in 1 2; do
for psd in 8 12 16 20 24 32 48 64; do
  timeout 300 python3 bench.py --welch --welch-psd $psd --steps 300 --warmup 30 2>/dev/null | tail -1 | \
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('psd $psd round $rnd: %.1f us per step, %.1f Gs/s, frac %.4f' % (d['ms_per_step']*1e3, d['value']/1e3, d['roofline']['frac']))"
done
done 2>&1 | tee $O/psd_sweep.txt

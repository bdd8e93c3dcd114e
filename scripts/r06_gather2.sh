#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/r06_gather2; mkdir -p $OUT
run() { tag=$1; shift; timeout 600 python bench.py --config c4 --gather-every-sweep --steps 20 --warmup 5 --settle 0.05 --no-cpu-baseline --no-overlap-leg --no-records-leg --no-hits-only-leg --no-copy-ref "$@" 2>>$OUT/err.txt > $OUT/$tag.json
  python - "$OUT/$tag.json" "$tag" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read()); g = d["gather_every_sweep"]
print(sys.argv[2], {k: g.get(k) for k in ("error", "sweeps_per_launch", "buffers_per_launch", "sweep_us", "sweep_with_gather_us", "exposed_gather_us", "exposed_frac", "gather_us", "host_us_per_sweep")}, (g.get("check") or {}).get("match"))
PY
}
run c16384 --centres 16384
run c8192 --centres 8192
run c2048_s4 --centres 2048
run c2048_s1 --centres 2048 --sweeps-per-launch 1
run c4096_s2 --centres 4096
bash scripts/r06_gather_probe.sh s4 2048 > $OUT/probe_s4.txt 2>&1; tail -60 $OUT/probe_s4.txt | cut -c1-170
timeout 900 python -m pytest tests/test_sweep_gpu.py tests/test_bench_contract_gpu.py tests/test_parity_gpu.py -m gpu -q -k "gather or window or bitmap or carries or c4" 2>&1 | tail -15

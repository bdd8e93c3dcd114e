#!/bin/bash
# Round 6: the steady-state gather on hardware (one rank: no peers) -- tests, then the C4 lines with the gather inside the timed region.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
OUT=gpurun_out/r06_gather
mkdir -p $OUT
export TMPDIR=/tmp
echo "== tests" > $OUT/tests.log
timeout 1500 python -m pytest tests/test_welch.py tests/test_sweep_gpu.py tests/test_bench_contract_gpu.py -m gpu -q -x 2>&1 | grep -v "^Frequency " | tail -40 >> $OUT/tests.log
for m in carry7 dcstale; do
  echo "== mutant $m" > $OUT/mutant_$m.log
  SCN_LIB=scanner_amd/variants/lib_$m.so timeout 900 python -m pytest tests/test_welch.py -m gpu -q 2>&1 | grep -v "^Frequency " | grep -E "FAILED|passed|failed" >> $OUT/mutant_$m.log
done
for c in 16384 2048; do
  timeout 900 python bench.py --config c4 --centres $c --gather-every-sweep --no-cpu-baseline --no-overlap-leg --no-records-leg --no-hits-only-leg --no-copy-ref \
     2>>$OUT/bench.err > $OUT/bench_c4_${c}_gather.json
done
python - <<'PY' | tee $OUT/summary.txt
import json
for c in (16384, 2048):
    try:
        d = json.loads(open(f"gpurun_out/r06_gather/bench_c4_{c}_gather.json").read())
        g = d["gather_every_sweep"]
        print(c, "value", d["value"], "| steady:", {k: g[k] for k in ("value", "value_without_gather", "sweep_us", "sweep_with_gather_us", "exposed_gather_us", "exposed_frac", "gather_us", "gather_us_min", "records_per_sweep", "root_blocked_in_wait_us_per_sweep")}, g["check"])
    except Exception as e:
        print(c, "failed:", e)
PY
tail -15 $OUT/tests.log; cat $OUT/mutant_*.log; tail -5 $OUT/bench.err

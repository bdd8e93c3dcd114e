#!/bin/bash
# Round 6: the steady-state gather on hardware (one rank: no peers), the two-consumer zero-copy host path, the whole GPU suite.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
OUT=gpurun_out/r06_gather
mkdir -p $OUT
export TMPDIR=/tmp
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
        print(c, "value", d["value"], "| steady:", {k: g.get(k) for k in ("error", "value", "value_without_gather", "sweep_us", "sweep_with_gather_us", "exposed_gather_us", "exposed_frac", "gather_us", "gather_us_min", "records_per_sweep", "root_blocked_in_wait_us_per_sweep")}, g.get("check"))
    except Exception as e:
        print(c, "failed:", e)
PY
echo "== host path" > $OUT/host_path.txt
timeout 900 bash scripts/host_path_rate.sh >> $OUT/host_path.txt 2>&1
echo "== full suite" > $OUT/suite.log
( time timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -v "^Frequency " | tail -60 ) >> $OUT/suite.log 2>&1
cat $OUT/host_path.txt; tail -25 $OUT/suite.log; tail -3 $OUT/bench.err

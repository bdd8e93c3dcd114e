#!/bin/bash
# Round 6: the GPU suite on the current tree, then the small-size steps (the total + trigger bitmap from the GPU, next 7).
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/r06_small; mkdir -p $OUT
( time timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -v "^Frequency " | tail -40 ) > $OUT/suite.log 2>&1
tail -12 $OUT/suite.log
for spec in "16 524288" "64 262144" "128 262144" "256 131072"; do set -- $spec
  timeout 600 python bench.py --n $1 --batch $2 --no-cpu-baseline --no-overlap-leg --no-records-leg --no-copy-ref --no-configs-leg 2>>$OUT/err.txt > $OUT/n$1.json
  python - $OUT/n$1.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read())
print(d["config"]["n"], d["config"]["buffers_per_launch"], "value", d["value"], "ms_per_step", d["ms_per_step"], "kernel_ms", d["roofline"]["kernel_avg_ms"], "hits_only", (d.get("hits_only") or {}).get("value"), (d.get("hits_only") or {}).get("ms_per_step"))
PY
done

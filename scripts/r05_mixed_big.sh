#!/bin/bash
# round 5: the mixed-radix kernels beyond 10000 points (two virtual threads per thread, one in-place exchange): parity, then time
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r05e; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_dispatch_gpu.py -x -q -k "12000 or 12288 or 14400 or 15000 or 16000 or documented" > $O/pytest_dispatch.txt 2>&1; echo "dispatch rc $?"; tail -3 $O/pytest_dispatch.txt
for cfg in "12000 2796" "12288 2730" "14400 2330" "15000 2236" "16000 2097"; do set -- $cfg
  for kind in cfloat int16; do
    timeout 300 python3 bench.py --n $1 --batch $2 --kind $kind --no-cpu-baseline --no-records-leg --no-overlap-leg --no-copy-ref --steps 300 --warmup 20 2>/dev/null | tail -1 >> $O/bench.jsonl
  done
done
python3 - <<'PY'
import json
for l in open("gpurun_out/r05e/bench.jsonl"):
    d = json.loads(l); r = d["roofline"]; h = d.get("hits_only") or {}
    print(d["config"]["n"], d["config"]["sample_kind"], d["config"]["buffers_per_launch"], "value %.1f Gs/s" % (d["value"]/1e3), "ms %.4f" % d["ms_per_step"], "frac", r["frac"], "hits-only %.1f" % (h.get("value",0)/1e3), r["kernel"][:64])
PY

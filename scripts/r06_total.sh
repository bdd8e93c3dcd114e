#!/bin/bash
# From how many buffers per launch does the total + trigger-bitmap reduction (on the launch's own stream, into pinned memory) pay?
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/r06_total; mkdir -p $OUT
timeout 600 python -m pytest tests/test_parity_gpu.py tests/test_oracle_ref_dsp.py -m gpu -q -k "bitmap or counts_by or live_reference or k1" 2>&1 | tail -4
for spec in "16 524288" "64 262144" "128 262144" "256 131072" "512 65536" "1024 32768"; do set -- $spec
  for v in product t17 t16 t15; do
    lib=""; [ $v != product ] && lib=scanner_amd/variants/lib_$v.so
    SCN_LIB=$lib timeout 600 python bench.py --n $1 --batch $2 --no-cpu-baseline --no-overlap-leg --no-records-leg --no-copy-ref --no-configs-leg 2>>$OUT/err.txt > $OUT/n$1_$v.json
    python - $OUT/n$1_$v.json $v <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read())
print(sys.argv[2], d["config"]["n"], d["config"]["buffers_per_launch"], "ms_per_step", d["ms_per_step"], "value", d["value"], "| hits-only ms", (d.get("hits_only") or {}).get("ms_per_step"), (d.get("hits_only") or {}).get("value"))
PY
  done
done | tee $OUT/table.txt

#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
( time timeout 400 python3 bench.py > $O/bench_r06_line.json 2> $O/bench_r06_line.err ) 2> $O/bench_r06_line_time.txt; grep real $O/bench_r06_line_time.txt
timeout 300 python3 bench.py --steps 20 --warmup 3 > $O/bench_r06_line_steps20.json 2>/dev/null
timeout 900 python3 -m pytest tests/test_host_cpp.py tests/test_bench_contract_gpu.py tests/test_oracle_ref_dsp.py -m gpu -q 2>&1 | grep -v "^Frequency " | tail -8
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_r06_line.json").read()); r = d["roofline"]
print(d["value"], d["ms_per_step"], r["frac"], r.get("frac_kernel_rocprof"), r.get("traffic"), r.get("traffic_stale"), r.get("traffic_build"))
PY

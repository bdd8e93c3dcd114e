#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r06_welch_double; mkdir -p $O
timeout 900 python3 -m pytest tests/test_welch.py -m gpu -q 2>&1 | grep -v "^Frequency " | tail -3
timeout 900 python3 scripts/r06_welch_acc.py 2>&1 | grep -v amdgpu.ids | tee $O/welch_acc.txt
for k in cfloat int16 int8; do timeout 300 python3 bench.py --welch --kind $k --steps 300 --warmup 20 2>/dev/null | tail -1; done > $O/welch.jsonl
timeout 300 python3 bench.py --welch --welch-psd 8 --steps 300 --warmup 20 2>/dev/null | tail -1 >> $O/welch.jsonl
python3 - <<'PY'
import json
for l in open("gpurun_out/r06_welch_double/welch.jsonl"):
    d = json.loads(l); print(d["config"]["kind"], d["config"]["psd_per_submit"], d["value"], d["ms_per_step"], d["c5_check"]["match"], d["c5_check"]["max_rel_power_vs_max_bin_mean"])
PY
SCN_PROF_KERNEL=scn_welch bash scripts/prof.sh r06d_c5 --welch > /dev/null 2>&1; grep -E "avg\(after|hbm_bytes_per_step|valu_frac" gpurun_out/prof_r06d_c5/summary.txt

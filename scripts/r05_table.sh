#!/bin/bash
# round 5: small-size steps with the frequency table resident on the GPU and the batch's total summed there
# (scn_plan_set_table / scn_submit_device_indexed / scn_hit_total_kernel) against per-buffer centres and the host's walk
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r05t; mkdir -p $O
timeout 900 python3 -m pytest tests/test_parity_gpu.py tests/test_sweep_gpu.py tests/test_bench_contract_gpu.py -q -k "indexed or total_of or alternating or sweep or gather or bench or contract" -x > $O/pytest.txt 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.txt
run() { tag=$1; shift; python3 bench.py --no-cpu-baseline --no-configs-leg "$@" 2>/dev/null | tail -1 > $O/line.json; python3 - "$tag" <<PY
import json,sys
d=json.load(open("$O/line.json")); print(f"{sys.argv[1]:34s} {d['value']/1e3:7.1f} Gs/s {d['ms_per_step']*1e3:7.1f} us  hits_only {(d.get('hits_only') or {}).get('value',0)/1e3:7.1f}")
PY
}
for shape in "16 524288" "64 262144" "128 262144" "256 131072" "512 65536" "1024 32768" "4096 8192"; do
  set -- $shape
  run "n=$1 per-buffer centres" --n $1 --batch $2 --per-buffer-centres
  SCN_LIB=scanner_amd/variants/lib_nototal.so run "n=$1 table, host walk" --n $1 --batch $2
  run "n=$1 table, GPU total" --n $1 --batch $2
done 2>&1 | tee $O/ab.txt

#!/bin/bash
# round 5: the mixed-radix fused kernels on the GPU -- parity first, then time
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r05b; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_dispatch_gpu.py -x -q -k "1000 or 1200 or 1500 or 2000 or 2400 or 2500 or 3000 or 3600 or 4000 or 4800 or 5000 or 6000 or 7200 or 8000 or 9000 or 9600 or documented" > $O/pytest_dispatch.txt 2>&1; echo "dispatch rc $?"; tail -3 $O/pytest_dispatch.txt
timeout 900 python3 -m pytest tests/test_parity_gpu.py -x -q -k "non_power_of_two or dc_quirk or alternating" > $O/pytest_parity.txt 2>&1; echo "parity rc $?"; tail -3 $O/pytest_parity.txt
for cfg in "1000 32768" "1500 22369" "3000 11184" "5000 6710" "6000 5592" "8000 4194" "10000 3355"; do set -- $cfg
  for kind in cfloat int16; do
    timeout 300 python3 bench.py --n $1 --batch $2 --kind $kind --no-cpu-baseline --no-records-leg --no-overlap-leg --no-copy-ref --steps 300 --warmup 20 2>/dev/null | tail -1 >> $O/bench.jsonl
  done
done
python3 - <<'PY'
import json
for l in open("gpurun_out/r05b/bench.jsonl"):
    d = json.loads(l); r = d["roofline"]; h = d.get("hits_only") or {}
    print(d["config"]["n"], d["config"]["sample_kind"], d["config"]["buffers_per_launch"], "value %.1f Gs/s" % (d["value"]/1e3), "ms %.4f" % d["ms_per_step"], "frac", r["frac"], "hits-only %.1f" % (h.get("value",0)/1e3), r["kernel"][:60])
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace1000 -- python3 bench.py --n 1000 --batch 32768 --no-cpu-baseline --no-records-leg --no-overlap-leg --no-hits-only-leg --no-copy-ref --steps 300 --warmup 20 > /dev/null 2>&1
grep -h "scn_fft_mixed" $O/trace1000/*/*kernel_stats.csv | head -3
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace6000 -- python3 bench.py --n 6000 --batch 5592 --no-cpu-baseline --no-records-leg --no-overlap-leg --no-hits-only-leg --no-copy-ref --steps 300 --warmup 20 > /dev/null 2>&1
grep -h "scn_fft_mixed" $O/trace6000/*/*kernel_stats.csv | head -3
rm -rf $O/trace1000 $O/trace6000

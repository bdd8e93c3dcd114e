#!/bin/bash
# per-kernel durations of the Welch step for variant libraries: rocprofv3 --kernel-trace --stats on bench.py --welch
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for v in "$@"; do
  d=gpurun_out/welch_kt_$v; rm -rf $d
  SCN_LIB=$PWD/scanner_amd/variants/lib_$v.so rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 bench.py --welch --welch-psd 32 --steps 200 --warmup 20 > /dev/null 2>&1
  echo "== $v"; python3 - <<PY
import csv, glob
for f in glob.glob("$d/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "welch" in r["Name"]: print("  ", r["Name"][:40], r["Calls"], round(float(r["AverageNs"])/1e3, 2), "us")
PY
done

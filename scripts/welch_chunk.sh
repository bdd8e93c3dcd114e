#!/bin/bash
# Welch: one measured attempt at keeping the work buffer Y in the Infinity Cache:
# a 32-PSD submit in chunks of c PSDs, chunk k's row kernel on a second stream beside chunk k+1's column kernel.
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/welch_chunk; mkdir -p $O
SCN_EXP_WELCH_CHUNK=4 timeout 600 python3 -m pytest tests/test_welch.py -x -q -m gpu 2>&1 | tail -1
for c in 0 2 4 8 16; do
  echo -n "chunk $c: "
  SCN_EXP_WELCH_CHUNK=$c python3 bench.py --welch --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value']/1e3, 'Gsamples/s,', d['ms_per_step']*1e3, 'us per step')"
done
for c in 0 4; do
  SCN_EXP_WELCH_CHUNK=$c rocprofv3 --kernel-trace --stats --output-format csv -d $O/t$c -- python3 bench.py --welch --steps 100 --warmup 10 --no-cpu-baseline > /dev/null 2> $O/t$c.log
  echo "chunk $c kernel stats:"; grep -h "scn_welch" $O/t$c/*/*kernel_stats.csv | cut -d, -f1-4
  for ctr in FETCH_SIZE WRITE_SIZE; do
    SCN_EXP_WELCH_CHUNK=$c rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/p${c}_$ctr -- python3 bench.py --welch --steps 20 --warmup 5 --settle 0 --no-cpu-baseline > /dev/null 2> $O/p${c}_$ctr.log
    python3 - $O/p${c}_$ctr $ctr <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        if "scn_welch" in row["Kernel_Name"]: acc[row["Kernel_Name"][:40]].append(float(row["Counter_Value"]))
print("   ", sys.argv[2], {k: round(sum(v) / len(v)) for k, v in acc.items()}, "(KiB per launch; FETCH_SIZE counts half on gfx950)")
PY
  done
  rm -rf $O/t$c $O/p${c}_*
done

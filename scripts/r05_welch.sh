#!/bin/bash
# round 5: the Welch column kernel keeps the overlapping half of a segment's input in registers -- A/B on one box
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r05f; mkdir -p $O
for v in default welch_nt; do
  lib=$PWD/scanner_amd/variants/lib_$v.so; [ $v = default ] && lib=$PWD/scanner_amd/libscanner_hip.so
  SCN_LIB=$lib timeout 600 python3 -m pytest tests/test_welch.py -x -q > $O/pytest_$v.txt 2>&1; echo "$v welch parity rc $?"; tail -2 $O/pytest_$v.txt
done
for rnd in 1 2 3; do
for v in welch_old default welch_nt; do
  lib=$PWD/scanner_amd/variants/lib_$v.so; [ $v = default ] && lib=$PWD/scanner_amd/libscanner_hip.so
  SCN_LIB=$lib timeout 300 python3 bench.py --welch --welch-psd 32 --steps 300 --warmup 30 2>/dev/null | tail -1 | \
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v round $rnd: %.1f us per step, %.1f Gs/s, frac %.4f' % (d['ms_per_step']*1e3, d['value']/1e3, d['roofline']['frac']))"
done
done 2>&1 | tee $O/ab.txt
for v in welch_old default; do
  lib=$PWD/scanner_amd/variants/lib_$v.so; [ $v = default ] && lib=$PWD/scanner_amd/libscanner_hip.so
  SCN_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$v -- python3 bench.py --welch --welch-psd 32 --steps 300 --warmup 30 > /dev/null 2>&1
  echo "$v:"; grep -h "scn_welch" $O/trace_$v/*/*kernel_stats.csv | head -3 | tee -a $O/ab.txt
  SCN_LIB=$lib rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "scn_welch" --kernel-trace --output-format csv -d $O/pmc_$v -- python3 bench.py --welch --welch-psd 32 --steps 40 --warmup 5 > /dev/null 2>&1
  python3 - $O/pmc_$v <<'PY' | tee -a gpurun_out/r05f/ab.txt
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        acc[row["Kernel_Name"][:30]].append(float(row["Counter_Value"]))
for k, v in acc.items(): print("  FETCH bytes per launch", k, "%.4g" % (sum(v) / len(v) * 2048))
PY
  rm -rf $O/trace_$v $O/pmc_$v
done

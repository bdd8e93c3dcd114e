#!/bin/bash
# kernel durations + VALU / wave-cycle counters of the three output modes of one launch shape:  scripts/mode_prof.sh <n> <kind> <batch>
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
N=$1; KIND=$2; NB=$3; O=gpurun_out/modeprof_${N}_${KIND}; mkdir -p $O
for f in 1 3 2; do
  python3 scripts/mode_loop.py $N $KIND $NB $f 300 | tail -1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/t$f -- python3 scripts/mode_loop.py $N $KIND $NB $f 300 > /dev/null 2> $O/t$f.log
  grep -h "scn_fft" $O/t$f/*/*kernel_stats.csv | head -2
  rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/p$f -- python3 scripts/mode_loop.py $N $KIND $NB $f 60 > /dev/null 2> $O/p$f.log
  python3 - $O/p$f <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        if "scn_fft" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
print("   per launch:", {k: round(sum(v) / len(v)) for k, v in sorted(acc.items())})
PY
  rm -rf $O/t$f $O/p$f
done

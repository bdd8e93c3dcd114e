#!/bin/bash
# Round evidence in one GPU call: rocprofv3 kernel trace + PMC passes for every BASELINE launch shape, un-profiled
# bench lines of the same build next to them.   usage: scripts/prof_all.sh <round tag>
TAG=${1:-r02}
cd "$GRAFT_REPO_ROOT"
bash scripts/prof.sh ${TAG}_c2 > /dev/null 2>&1
bash scripts/prof.sh ${TAG}_c3 --n 8192 --kind int16 --batch 4096 > /dev/null 2>&1
bash scripts/prof.sh ${TAG}_c4shape --batch 2048 > /dev/null 2>&1
SCN_PROF_KERNEL=scn_welch bash scripts/prof.sh ${TAG}_c5 --welch > /dev/null 2>&1
for c in c2 c3 c4shape c5; do echo "=== $c"; cat gpurun_out/prof_${TAG}_$c/summary.txt; done

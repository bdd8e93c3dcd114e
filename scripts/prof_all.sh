#!/bin/bash
# Round evidence in one GPU call: rocprofv3 kernel trace + PMC passes for every BASELINE launch shape, un-profiled
# bench lines of the same build next to them.   usage: scripts/prof_all.sh <round tag>
TAG=${1:-r03}
cd "$GRAFT_REPO_ROOT"
bash scripts/prof.sh ${TAG}_c1 --n 1024 --batch 32768 > /dev/null 2>&1
bash scripts/prof.sh ${TAG}_c2 > /dev/null 2>&1
bash scripts/prof.sh ${TAG}_c3 --n 8192 --kind int16 --batch 4096 > /dev/null 2>&1
bash scripts/prof.sh ${TAG}_c4shape --batch 2048 > /dev/null 2>&1
SCN_PROF_KERNEL=scn_welch bash scripts/prof.sh ${TAG}_c5 --welch > /dev/null 2>&1
# what the product's own worker runs (ProcessSamples::ThreadWorker creates plans without SCN_OUT_SPECTRUM, scanner_amd/host/process.cpp;
# the reference's defaults: 8192 points on int16 or int8, scan.cpp:85,138-140,183; time-domain mode is the CLI's default, scan.cpp:87)
bash scripts/prof.sh ${TAG}_c2_hitsonly --plan-mode hits > /dev/null 2>&1
bash scripts/prof.sh ${TAG}_n8192int16_hitsonly --n 8192 --kind int16 --batch 4096 --plan-mode hits > /dev/null 2>&1
bash scripts/prof.sh ${TAG}_n8192int8_hitsonly --n 8192 --kind int8 --batch 4096 --plan-mode hits > /dev/null 2>&1
bash scripts/prof.sh ${TAG}_n4096int16_hitsonly --kind int16 --plan-mode hits > /dev/null 2>&1
bash scripts/prof.sh ${TAG}_n8192int16_hitsonly_dc --n 8192 --kind int16 --batch 4096 --plan-mode hits --dc > /dev/null 2>&1
bash scripts/prof.sh ${TAG}_n8192int16p_hitsonly --n 8192 --kind int16p --batch 4096 --plan-mode hits > /dev/null 2>&1
SCN_PROF_KERNEL=scn_time_domain bash scripts/prof.sh ${TAG}_n8192int16_td --n 8192 --kind int16 --batch 4096 --time-domain > /dev/null 2>&1
if [ -n "$SCN_PROF_MORE" ]; then   # the other wire formats and sizes (not BASELINE launch shapes)
  bash scripts/prof.sh ${TAG}_n4096int16 --kind int16 > /dev/null 2>&1
  bash scripts/prof.sh ${TAG}_n4096int8 --kind int8 > /dev/null 2>&1
  bash scripts/prof.sh ${TAG}_n8192cfloat --n 8192 --batch 4096 > /dev/null 2>&1
  bash scripts/prof.sh ${TAG}_n16384cfloat --n 16384 --batch 2048 > /dev/null 2>&1
  bash scripts/prof.sh ${TAG}_n16384int16 --n 16384 --batch 2048 --kind int16 > /dev/null 2>&1
  bash scripts/prof.sh ${TAG}_n512cfloat --n 512 --batch 65536 > /dev/null 2>&1
  bash scripts/prof.sh ${TAG}_n256cfloat --n 256 --batch 131072 > /dev/null 2>&1
  bash scripts/prof.sh ${TAG}_n128cfloat --n 128 --batch 262144 > /dev/null 2>&1
  bash scripts/prof.sh ${TAG}_n64cfloat --n 64 --batch 262144 > /dev/null 2>&1
  bash scripts/prof.sh ${TAG}_n16cfloat --n 16 --batch 524288 > /dev/null 2>&1
  bash scripts/prof.sh ${TAG}_n1000cfloat --n 1000 --batch 32768 > /dev/null 2>&1      # the mixed-radix fused kernels (scn_mixed.hip)
  bash scripts/prof.sh ${TAG}_n6000int16 --n 6000 --batch 5592 --kind int16 > /dev/null 2>&1
  bash scripts/prof.sh ${TAG}_n10000cfloat --n 10000 --batch 3355 > /dev/null 2>&1
  SCN_PROF_KERNEL=scn_big bash scripts/prof.sh ${TAG}_n65536cfloat --n 65536 --batch 512 > /dev/null 2>&1
  SCN_PROF_KERNEL=scn_big bash scripts/prof.sh ${TAG}_n32768cfloat --n 32768 --batch 1024 > /dev/null 2>&1
fi
for c in c1 c2 c3 c4shape c5 c2_hitsonly n8192int16_hitsonly n8192int8_hitsonly n4096int16_hitsonly n8192int16_hitsonly_dc n8192int16p_hitsonly n8192int16_td; do echo "=== $c"; cat gpurun_out/prof_${TAG}_$c/summary.txt; done

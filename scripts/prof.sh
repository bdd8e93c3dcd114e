#!/bin/bash
# Profiles bench.py on the GPU box: kernel trace/stats, then PMC passes (separate runs).
# usage: scripts/prof.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
BENCH_ARGS="--steps 500 --warmup 5 --no-cpu-baseline --no-overlap-leg --no-records-leg --no-hits-only-leg --no-copy-ref --no-configs-leg $*"   # the contract leg only: overlapped kernels have inflated begin-to-end times
PMC_ARGS="--steps 50 --warmup 5 --settle 0 --no-cpu-baseline --no-overlap-leg --no-records-leg --no-hits-only-leg --no-copy-ref --no-configs-leg $*"   # counters serialise the launches: a short run is enough
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py $BENCH_ARGS > "$OUT/bench_trace.json" 2> "$OUT/trace.log"
for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
            "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" \
            "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  # (counters for this library's kernels only: at 65536-buffer batches the input generator's thousands of torch launches under
  # counter collection crashed rocprofv3 in round 3)
  rocprofv3 --pmc $pass --kernel-include-regex "scn_" --kernel-trace --output-format csv -d "$OUT/pmc_$name" -- python3 bench.py $PMC_ARGS > /dev/null 2> "$OUT/pmc_$name.log"
done
python3 scripts/prof_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
# keep only what is judged (the raw traces are tens of MB)
cp "$OUT"/trace/*/*kernel_stats.csv "$OUT/kernel_stats.csv" 2>/dev/null
rm -rf "$OUT"/trace "$OUT"/pmc_*/ 2>/dev/null
cat "$OUT/summary.txt"

#!/bin/bash
# Round 6, after the evidence run: (1) the single-kernel C5 skeleton against the pair's 127 us; (2) the gather loop from a C++ caller
# on the system HIP runtime (abi_bench --mode gather) beside its plain twin, at the 8-GPU share (2048 buffers per launch) and at 8192.
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r06_extra; mkdir -p $O
{
for a in "32 1" "32 2" "32 4" "64 1" "64 2" "16 4"; do ./scripts/ubench/c5_single $a 20; done
} 2>&1 | tee $O/c5_single.txt
{
for b in 2048 8192; do
  fl=0; [ $b = 2048 ] && fl=4
  for hits in 0 1; do
    ./scanner_amd/host/abi_bench --batch $b --flags $fl --depth 4 --mode counts --lag 2 --steps 2000 --hits-only $hits 2>/dev/null
    ./scanner_amd/host/abi_bench --batch $b --flags $fl --depth 4 --mode gather --steps 2000 --hits-only $hits 2>/dev/null
  done
done
} | tee $O/abi_gather.jsonl | cut -c1-330
timeout 900 python3 scripts/r06_welch_acc.py 2>&1 | grep -v amdgpu.ids | tee $O/welch_acc.txt

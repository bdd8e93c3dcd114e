#!/bin/bash
# round 6 evidence, one GPU call: full GPU suite, the mutation checks, rocprofv3 + PMC for every launch shape on THIS build, the bench lines,
# the Welch wire formats, the C4 steady-state legs with the gather in the loop, the other configs, the sweep table, ten minutes of fuzz
# (sample rates and Welch plans drawn), the host path rate with one and two consumers
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
TAG=${1:-r06}
O=gpurun_out; mkdir -p $O
timeout 1800 python3 -m pytest tests -m gpu -q > $O/pytest_$TAG.txt 2>&1; echo "pytest rc $?"; tail -3 $O/pytest_$TAG.txt
for m in carry7 dcstale; do
  if [ -f scanner_amd/variants/lib_$m.so ]; then
    SCN_LIB=scanner_amd/variants/lib_$m.so timeout 900 python3 -m pytest tests/test_welch.py -m gpu -q 2>&1 | grep -v "^Frequency " | grep -E "FAILED|passed|failed" > $O/mutant_${m}_$TAG.txt; tail -1 $O/mutant_${m}_$TAG.txt
  fi
done
SCN_PROF_MORE=1 timeout 4500 bash scripts/prof_all.sh $TAG > $O/prof_all_$TAG.txt 2>&1; echo "prof_all rc $?"
SCN_PROF_KERNEL=scn_welch bash scripts/prof.sh ${TAG}_c5_int16 --welch --kind int16 > /dev/null 2>&1
SCN_PROF_KERNEL=scn_welch bash scripts/prof.sh ${TAG}_c5_int16_dc --welch --kind int16 --dc > /dev/null 2>&1
SCN_PROF_KERNEL=scn_welch bash scripts/prof.sh ${TAG}_c5_int8 --welch --kind int8 > /dev/null 2>&1
( time timeout 400 python3 bench.py > $O/bench_${TAG}_final.json 2> $O/bench_${TAG}_final.err ) 2> $O/bench_${TAG}_time.txt; echo "bench rc $?"; grep real $O/bench_${TAG}_time.txt
timeout 300 python3 bench.py --steps 20 --warmup 3 > $O/bench_${TAG}_steps20.json 2>/dev/null; echo "bench20 rc $?"
for k in cfloat int16 int16p int8; do timeout 300 python3 bench.py --welch --kind $k --steps 200 --warmup 20 2>/dev/null | tail -1; done > $O/welch_kinds_$TAG.jsonl
timeout 300 python3 bench.py --welch --kind int16 --dc --steps 200 --warmup 20 2>/dev/null | tail -1 >> $O/welch_kinds_$TAG.jsonl
timeout 300 python3 bench.py --welch --welch-pinned --steps 30 --warmup 5 2>/dev/null | tail -1 >> $O/welch_kinds_$TAG.jsonl
timeout 300 python3 bench.py --welch --welch-pinned --kind int16 --steps 30 --warmup 5 2>/dev/null | tail -1 >> $O/welch_kinds_$TAG.jsonl
for c in 16384 8192 4096 2048; do timeout 400 python3 bench.py --config c4 --centres $c --gather-every-sweep --no-cpu-baseline --no-overlap-leg --no-records-leg --no-hits-only-leg --no-copy-ref 2>/dev/null | tail -1; done > $O/c4_gather_$TAG.jsonl
timeout 400 python3 bench.py --config c4 --centres 2048 --sweeps-per-launch 1 --gather-every-sweep --no-cpu-baseline --no-overlap-leg --no-records-leg --no-hits-only-leg --no-copy-ref 2>/dev/null | tail -1 >> $O/c4_gather_$TAG.jsonl
bash scripts/r06_gather_probe.sh ${TAG}_s4 2048 > $O/gather_probe_$TAG.txt 2>&1
timeout 1800 bash scripts/other_configs.sh $TAG > $O/other_$TAG.txt 2>&1; echo "other rc $?"
timeout 900 python3 scripts/sweep_all.py 256 512 1024 2048 4096 8192 16384 > $O/sweep_all_$TAG.txt 2>&1; echo "sweep rc $?"
timeout 700 python3 scripts/fuzz_parity.py 600 61 > $O/fuzz_$TAG.txt 2>&1; echo "fuzz rc $?"; tail -5 $O/fuzz_$TAG.txt
bash scripts/host_path_rate.sh > $O/host_path_rate_$TAG.txt 2>&1; echo "host path rc $?"

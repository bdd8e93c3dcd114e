#!/bin/bash
# Round 6, first GPU call: the new Welch / sample-rate tests, the mutation checks, then the whole GPU suite and the bench lines.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
OUT=gpurun_out/r06_first
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import torch; print(torch.cuda.get_device_name(0))" > $OUT/device.txt 2>&1
echo "== new tests" | tee $OUT/new_tests.log
timeout 1500 python -m pytest tests/test_welch.py tests/test_sample_rates_gpu.py -m gpu -q -x -s 2>&1 | grep -v "^Frequency " | tail -80 >> $OUT/new_tests.log
echo "rc=$?" >> $OUT/new_tests.log
# mutation checks: the same tests against libraries with one deliberate fault each must FAIL
for m in carry7 dcstale; do
  echo "== mutant $m" > $OUT/mutant_$m.log
  SCN_LIB=scanner_amd/variants/lib_$m.so timeout 900 python -m pytest tests/test_welch.py -m gpu -q 2>&1 | grep -v "^Frequency " | tail -40 >> $OUT/mutant_$m.log
done
echo "== bench" > $OUT/bench.log
timeout 900 python bench.py 2>$OUT/bench.err | tee $OUT/bench_default.json | cut -c1-600 >> $OUT/bench.log
timeout 600 python bench.py --welch --steps 100 --warmup 10 2>>$OUT/bench.err | tee $OUT/bench_welch.json | cut -c1-1500 >> $OUT/bench.log
timeout 600 python bench.py --welch --kind int16 --steps 100 --warmup 10 2>>$OUT/bench.err | tee $OUT/bench_welch_int16.json | cut -c1-1500 >> $OUT/bench.log
timeout 600 python bench.py --welch --kind int16 --dc --steps 100 --warmup 10 2>>$OUT/bench.err | tee $OUT/bench_welch_int16_dc.json | cut -c1-1500 >> $OUT/bench.log
timeout 600 python bench.py --welch --kind int8 --steps 100 --warmup 10 2>>$OUT/bench.err | tee $OUT/bench_welch_int8.json | cut -c1-1500 >> $OUT/bench.log
timeout 600 python bench.py --welch --welch-pinned --steps 30 --warmup 5 2>>$OUT/bench.err | tee $OUT/bench_welch_pinned.json | cut -c1-1500 >> $OUT/bench.log
echo "== full suite" > $OUT/suite.log
( time timeout 1700 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "^Frequency " | tail -30 ) >> $OUT/suite.log 2>&1
tail -5 $OUT/new_tests.log; tail -4 $OUT/mutant_carry7.log; tail -4 $OUT/mutant_dcstale.log; tail -8 $OUT/suite.log; cat $OUT/bench.log | cut -c1-400

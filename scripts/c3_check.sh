python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "8192 or c3 or sizes" 2>&1 | grep -E "passed|failed|Error" | tail -3
for i in 1 2; do python3 scripts/loop_only.py 1500 0 8192 int16 4096 2>/dev/null | tail -1; python3 scripts/loop_only.py 1500 0 8192 cfloat 4096 2>/dev/null | tail -1; done
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --kernel-trace --output-format csv -d gpurun_out/pmc_lds -- python3 scripts/loop_only.py 60 0 8192 int16 4096 > /dev/null 2>&1
python3 - <<'PY'
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_lds/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "fft8k" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items(): print(k, sum(v)/len(v), len(v))
PY
rm -rf gpurun_out/pmc_lds

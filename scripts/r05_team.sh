#!/bin/bash
# round 5: the L2-resident work buffer, decided by its memory skeleton (scripts/ubench/team_l2.hip)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r05c; mkdir -p $O
for tpx in 2 3 4 6; do timeout 60 scripts/ubench/team_l2 512 $tpx 20; done 2>&1 | tee $O/team_l2.txt
timeout 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -- scripts/ubench/team_l2 512 4 10 > /dev/null 2>&1
timeout 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -- scripts/ubench/team_l2 512 4 10 > /dev/null 2>&1
python3 - <<'PY' | tee -a gpurun_out/r05c/team_l2.txt
import csv, glob, collections
for tag, scale in (("f", 2048), ("w", 1024)):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/r05c/pmc_{tag}/*/*counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            acc[(row["Kernel_Name"][:40], row["Counter_Name"])].append(float(row["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print(k, "avg bytes per launch %.4g" % (sum(v) / len(v) * scale), "n", len(v))
PY
rm -rf $O/pmc_f $O/pmc_w

"""Summarise a scripts/prof.sh output directory: per-kernel stats + per-launch PMC averages."""
def _source_hash():
    """the build these counters were taken on (scanner_amd.build.source_hash): bench.py only reports them for the same one"""
    try:
        import os as _os, sys as _sys
        _sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
        from scanner_amd import build
        return build.source_hash()
    except Exception:
        return None


import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]
KER = os.environ.get("SCN_PROF_KERNEL", "scn_fft")  # substring of the kernel(s) being judged ("scn_welch" for C5)


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f in find("trace/**/*kernel_stats.csv"):
    for row in csv.DictReader(open(f)):
        d = {k: row[k] for k in row if k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")}
        d["Name"] = d.get("Name", "")[:90]
        if float(d.get("Percentage", 0) or 0) >= 0.5:
            print(d)
recs = []
for f in find("trace/**/*kernel_trace.csv"):
    for row in csv.DictReader(open(f)):
        if KER in row.get("Kernel_Name", ""):
            recs.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]) - int(row["Start_Timestamp"])))
            meta = {k: row.get(k) for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size",
                                            "Workgroup_Size_X", "Grid_Size_X")}


def stats(label, d):
    d = sorted(d)
    print(f"{label}: n={len(d)} avg={sum(d)/len(d)/1e3:.2f}us median={d[len(d)//2]/1e3:.2f}us "
          f"min={d[0]/1e3:.2f}us max={d[-1]/1e3:.2f}us")


by_name = defaultdict(list)
for f in find("trace/**/*kernel_trace.csv"):
    for row in csv.DictReader(open(f)):
        if KER in row.get("Kernel_Name", ""):
            by_name[row["Kernel_Name"][:70]].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
kernel_avg_us = {}
for name, d in by_name.items():
    tail = d[len(d) // 5:]  # drop the settle / warm-up fifth
    kernel_avg_us[name] = sum(tail) / len(tail) / 1e3
    print(f"  {name}: n={len(d)} avg(after first 20%)={kernel_avg_us[name]:.2f}us")
if recs:
    recs.sort()
    print(f"{KER} launches in the trace (settle + warmup + timed steps + the final hit-list sweep):", meta)
    stats("  all launches", [r[1] for r in recs])
    try:  # the timed region of bench.py = the `steps` launches before the final sweep's one
        steps = json.loads(open(os.path.join(out, "bench_trace.json")).read().strip().splitlines()[-1])["steps"]
        stats(f"  the {steps} launches of bench.py's timed region", [r[1] for r in recs[-(steps + 1):-1]])
    except Exception as e:  # pragma: no cover
        print("  (timed region not separable:", e, ")")
print("== PMC (average per launch of the FFT kernel) ==")
summary = {}
per_kernel = defaultdict(dict)
for f in find("pmc_*/**/*counter_collection.csv"):
    acc = defaultdict(list)
    acck = defaultdict(lambda: defaultdict(list))
    for row in csv.DictReader(open(f)):
        if KER in row.get("Kernel_Name", ""):
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
            acck[row["Kernel_Name"][:70]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        summary[k] = sum(v) / len(v)
        print(f"{k:28s} {summary[k]:.4g}   (n={len(v)})")
    for kn, cs in acck.items():
        for k, v in cs.items():
            per_kernel[kn][k] = sum(v) / len(v)
if len(per_kernel) > 1:
    for kn, cs in per_kernel.items():
        if "FETCH_SIZE" in cs or "WRITE_SIZE" in cs:
            print(f"  {kn}: read {cs.get('FETCH_SIZE', 0) * 2048:.4g} B  write {cs.get('WRITE_SIZE', 0) * 1024:.4g} B per launch")
if "FETCH_SIZE" in summary or "WRITE_SIZE" in summary:
    # MI355X_MICROARCH.md (HBM): FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports
    # exactly half of the bytes of a wide coalesced streaming read -> double it.
    fetch = summary.get("FETCH_SIZE", 0) * 1024 * 2
    write = summary.get("WRITE_SIZE", 0) * 1024
    print(f"hbm_bytes_per_launch (FETCH_SIZE*1024*2 + WRITE_SIZE*1024) = {fetch + write:.4g}  (read {fetch:.4g}, write {write:.4g})")
    shape = {}
    try:  # the launch shape the counters belong to (bench.py only reports them for the same shape)
        cfg = json.loads(open(os.path.join(out, "bench_trace.json")).read().strip().splitlines()[-1])["config"]
        shape = {k: cfg.get(k) for k in ("n", "sample_kind", "batch_per_gpu", "buffers_per_launch", "segments_per_psd", "psd_per_submit",
                                         "plan_mode", "correct_dc", "time_domain")}
    except Exception:
        pass
    if len(per_kernel) > 1:  # several kernels make up one step (Welch: columns + rows): bytes per step = the sum
        fetch = sum(cs.get("FETCH_SIZE", 0) for cs in per_kernel.values()) * 1024 * 2
        write = sum(cs.get("WRITE_SIZE", 0) for cs in per_kernel.values()) * 1024
        print(f"hbm_bytes_per_step over {len(per_kernel)} kernels = {fetch + write:.4g}  (read {fetch:.4g}, write {write:.4g})")
    # the second bound, for launches that are not HBM-bound: the share of the chip's VALU issue slots the launch used.  SQ_INSTS_VALU
    # counts wave instructions; a 64-lane wave instruction occupies its 16-lane SIMD for 4 cycles (double-precision ones longer: a
    # lower bound there); 256 CUs x 4 SIMDs at the 2.4 GHz peak clock (MI355X_MICROARCH.md)
    k_us = sum(kernel_avg_us.values()) if kernel_avg_us else None
    valu = summary.get("SQ_INSTS_VALU")
    valu_frac = (valu * 4.0 / (k_us * 1e-6 * 2.4e9 * 1024)) if (valu and k_us) else None
    if valu_frac is not None:
        print(f"valu_frac = SQ_INSTS_VALU {valu:.4g} x 4 cycles / ({k_us:.2f} us x 2.4 GHz x 1024 SIMDs) = {valu_frac:.3f}")
    json.dump(dict({"hbm_bytes_per_launch": fetch + write, "read_bytes": fetch, "write_bytes": write,
                    "insts_valu": valu, "valu_frac": None if valu_frac is None else round(valu_frac, 4),
                    "kernel_avg_us": sum(kernel_avg_us.values()) if kernel_avg_us else None,
                    "kernels": {k: round(v, 3) for k, v in kernel_avg_us.items()},
                    "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; KiB units; FETCH_SIZE x2 (gfx950 correction)",
                    "build": _source_hash()},
                   **shape),
              open(os.path.join(out, "pmc_traffic.json"), "w"))
try:  # (the figures of the traced run itself -- not the roofline object's tracked-profile fields, which at trace time still describe the build BEFORE this one)
    line = json.loads(open(os.path.join(out, "bench_trace.json")).read().strip().splitlines()[-1])
    r = line.get("roofline") or {}
    print("bench line under trace:", json.dumps({"value": line.get("value"), "unit": line.get("unit"), "steps": line.get("steps"), "ms_per_step": line.get("ms_per_step"),
                                                  "kernel_avg_ms_hip_events": r.get("kernel_avg_ms"), "frac_event": r.get("frac"),
                                                  "workload": (line.get("config") or {}).get("workload", "")[:160]}))
except Exception:
    pass

"""Copy one round's profiling evidence from gpurun_out/ (scratch) into profiles/ (tracked):
   python scripts/publish_profiles.py r02
summary.txt -> profiles/<tag>_<shape>_rocprofv3_summary.txt, kernel_stats.csv beside it, and the per-shape PMC
traffic / kernel averages into profiles/measured_shapes.json (what bench.py reads for roofline.traffic)."""
import json, os, shutil, sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scanner_amd import build as _build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
shapes_path = os.path.join(ROOT, "profiles", "measured_shapes.json")
shapes = json.load(open(shapes_path)) if os.path.exists(shapes_path) else {}
import glob

found = sorted(os.path.basename(d)[len(f"prof_{tag}_"):] for d in glob.glob(os.path.join(ROOT, "gpurun_out", f"prof_{tag}_*")))
for shape in found:  # every launch shape scripts/prof_all.sh profiled under this tag
    d = os.path.join(ROOT, "gpurun_out", f"prof_{tag}_{shape}")
    if not os.path.exists(os.path.join(d, "summary.txt")):
        print("missing", d)
        continue
    shutil.copy(os.path.join(d, "summary.txt"), os.path.join(ROOT, "profiles", f"{tag}_{shape}_rocprofv3_summary.txt"))
    shutil.copy(os.path.join(d, "kernel_stats.csv"), os.path.join(ROOT, "profiles", f"{tag}_{shape}_kernel_stats.csv"))
    if not os.path.exists(os.path.join(d, "pmc_traffic.json")):  # (a counter pass that did not finish: the summary says so)
        print("no traffic counters for", shape)
        continue
    t = json.load(open(os.path.join(d, "pmc_traffic.json")))
    welch = t.get("segments_per_psd") is not None
    mode = t.get("plan_mode") or "both"
    key = (f"welch/{t['n']}/{t['segments_per_psd']}/{t['psd_per_submit']}" + ("" if (t.get("sample_kind") or "cfloat") == "cfloat" else f"/{t['sample_kind']}") +
           ("/dc" if t.get("correct_dc") else "") if welch
           else f"{t['n']}/{t['sample_kind']}/{t['buffers_per_launch']}" +  # = bench.py's shape_key
                ("/td" if t.get("time_domain") else "" if mode == "both" else "/" + mode) + ("/dc" if t.get("correct_dc") else ""))
    entry = {("hbm_bytes_per_step" if welch else "hbm_bytes_per_launch"): int(round(t["hbm_bytes_per_launch"])),
             "read_bytes": int(round(t["read_bytes"])), "write_bytes": int(round(t["write_bytes"])),
             "kernel_avg_us": round(t["kernel_avg_us"], 3), "kernels": t["kernels"],
             "insts_valu": t.get("insts_valu"), "valu_frac": t.get("valu_frac"),
             "source": f"profiles/{tag}_{shape}_rocprofv3_summary.txt",
             # which build the counters were taken on (the profiling run writes it; the current tree's hash otherwise)
             "build": t.get("build") or _build.source_hash(),
             "method": t["method"] + ("; summed over the kernels of one step" if welch else "")}
    shapes[key] = entry
    print(key, entry["kernel_avg_us"], "us")
json.dump(shapes, open(shapes_path, "w"), indent=1)
for src, dst in ((f"configs_{tag}.jsonl", f"{tag}_other_configs.jsonl"), (f"other_{tag}.txt", f"{tag}_other_configs.txt"),
                 (f"bench_{tag}_final.json", f"{tag}_bench_line.json")):
    p = os.path.join(ROOT, "gpurun_out", src)
    if os.path.exists(p) and os.path.getsize(p):
        shutil.copy(p, os.path.join(ROOT, "profiles", dst))

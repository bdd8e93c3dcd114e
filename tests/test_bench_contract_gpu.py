"""bench.py prints ONE JSON line carrying the driver's contract fields plus `roofline`, `cpu_baseline`
and (extra) `overlap`; checked end to end on the GPU box with a tiny K/W, single process and through
torch.distributed with one rank (the N > 1 code path: RCCL init, barrier, max-over-ranks, gather)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CONTRACT = {"metric": str, "value": (int, float), "unit": str, "n_gpus": int, "steps": int, "warmup": int,
            "ms_per_step": (int, float), "higher_is_better": bool, "scaling": str, "dtype": str, "data": str,
            "config": dict}


def run_bench(extra, env=None):
    e = dict(os.environ)
    e.update(env or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "2", "--settle",
                          "0.05", "--batch", "512", "--cpu-seconds", "0.6"] + extra,
                         capture_output=True, text=True, timeout=600, env=e, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, "bench.py must print exactly one line on stdout"
    return json.loads(lines[0])


def check_common(d, steps=6, warmup=2):
    for k, t in CONTRACT.items():
        assert k in d and isinstance(d[k], t), k
    assert "vs_baseline" in d and d["vs_baseline"] is None          # BASELINE.md publishes no number for this metric
    assert d["steps"] == steps and d["warmup"] == warmup and d["n_gpus"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["data"] == "synthetic"
    assert d["unit"] == "Msamples/s" and d["value"] > 0 and d["ms_per_step"] > 0
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    # value and ms_per_step describe the same run
    samples = d["config"]["batch_per_gpu"] * d["config"]["n"]
    assert abs(d["value"] - samples / (d["ms_per_step"] * 1e-3) / 1e6) / d["value"] < 0.02


def test_bench_line_single_process():
    d = run_bench([])
    check_common(d)
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "Msamples/s" and c["value"] > 0 and c["cores"] >= 1 and c["sample"]
    o = d["overlap"]
    assert o["unit"] == "Msamples/s" and o["value"] > 0
    assert d["settle_steps"] >= 2 and "settle" in d["config"]["workload"]
    r = d["with_hit_records"]
    # the records legs: from a C++ child on the system HIP runtime (value), and from this torch process beside it
    assert r["value"] > 0 and r["hits_per_step"] > 0 and "scn_hits_view" in r["path"] and r["hip_runtime_version"] > 0
    assert r["copied_out_by_scn_collect"]["value"] > 0 and r["two_in_flight"]["value"] > 0
    assert r["hits_only_plan"]["value"] > 0 and r["hits_only_plan"]["four_in_flight"]["value"] > 0   # the worker's kind of plan
    py = r["python_torch_runtime"]
    assert py["value"] > 0 and py["collect_hits"] > 0 and py["collect_with_records_us"] > 0
    assert d["roofline"]["kernel"].startswith("scn_fft_kernel<16, SCN_K_FLOAT_COMPLEX")
    assert d["roofline"]["frac_wall"] <= d["roofline"]["frac_event"] * 1.05
    h = d["hits_only"]                                   # hits-only mode is reported separately, on its own byte count
    assert h["value"] > 0 and h["algorithmic_bytes_per_sample"] == 8 and h["plan_flags"] == "SCN_OUT_HITS"
    assert d["roofline"]["measured_copy_GBs"] > 500      # a device-to-device copy measured in the same run


def test_bench_line_through_torch_distributed_one_rank():
    d = run_bench(["--no-cpu-baseline", "--no-overlap-leg", "--kind", "int16", "--n", "8192"],
                  env={"SCN_BENCH_FORCE_DIST": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29517",
                       "RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1"})
    check_common(d)
    assert d["cpu_baseline"] is None and d["overlap"] is None
    assert d["config"]["sample_kind"] == "int16" and d["roofline"]["algorithmic_bytes_per_sample"] == 8
    assert d["final_sweep_hits"] >= 0
    assert d["roofline"]["kernel"].startswith("scn_fft8k_kernel<SCN_K_SHORT_COMPLEX")      # the kernel that actually ran
    g = d["gather"]                                                                        # the C-ABI's RCCL gather, not torch's
    assert g["transport"].startswith("scn_gather_hits") and g["globally_ordered"] is True
    assert g["per_rank"] == [d["final_sweep_hits"]]

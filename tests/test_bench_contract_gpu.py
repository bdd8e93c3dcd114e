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


def run_bench(extra, env=None, batch=("--batch", "512")):
    e = dict(os.environ)
    e.update(env or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "2", "--settle",
                          "0.05", *batch, "--cpu-seconds", "0.6"] + extra,
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
    # ... with the kernel the worker's plans run, by name, and its own clocks / counters like `roofline` has
    assert h["kernel"] == "scn_fft_kernel<16, SCN_K_FLOAT_COMPLEX, false, true, false>" and h["kernel_avg_ms"] > 0
    assert abs(h["frac"] - h["algorithmic_bytes_per_launch"] / (h["kernel_avg_ms"] * 1e-3) / 1e9 / 8000.0) < 1e-3
    for k in ("frac_kernel_rocprof", "traffic", "valu_frac", "traffic_source", "traffic_build", "traffic_stale"):
        assert k in h, k
    assert "valu_frac" in d["roofline"] and "frac_kernel_rocprof" in d["roofline"]
    assert "configs" not in d                            # (they ride on the default C2 line only: this run has --batch 512)
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


def test_default_multi_rank_line_carries_the_strong_scaling_sweep():
    """`bench.py --gpus N` (N > 1) prints the weak-scaling value AND configs.c4: BASELINE config 4 sharded over the ranks (strong
    scaling) with every launch's hit list gathered inside the timed region.  Here through torch.distributed with ONE rank (the
    N > 1 code path: process group, barriers, RCCL communicator)."""
    d = run_bench(["--no-cpu-baseline", "--no-overlap-leg", "--no-records-leg", "--no-hits-only-leg", "--no-copy-ref"], batch=(),
                  env={"SCN_BENCH_FORCE_DIST": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29519",
                       "RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1"})
    check_common(d)
    assert d["scaling"] == "weak" and d["config"]["buffers_per_launch"] == 8192
    g = d["configs"]["c4"]
    check_gather_leg(g, centres=16384, per_gpu=16384)
    assert g["launches_per_sweep"] == 2 and "16384 centres" in g["workload"]


def test_default_line_carries_a_leg_per_baseline_config():
    """`python bench.py` (C2) also times C3, the C4 per-GPU share and C5 on the same box in the same run"""
    d = run_bench(["--no-records-leg", "--no-overlap-leg", "--no-copy-ref"], batch=())
    check_common(d)
    assert d["config"]["buffers_per_launch"] == 8192 and d["config"]["workload"].startswith("C2:")
    c = d["configs"]
    assert "error" not in c, c
    for leg, kern in (("c3", "scn_fft8k_kernel<SCN_K_SHORT_COMPLEX, false, true, true>"),
                      ("c4_share", "scn_fft_kernel<16, SCN_K_FLOAT_COMPLEX, false, true, true>"), ("c5", "scn_welch_cols_kernel")):
        g = c[leg]
        assert g["value"] > 0 and g["ms_per_step"] > 0 and 0 < g["frac"] < 1 and g["kernel"].startswith(kern) and g["workload"]
        for k in ("frac_kernel_rocprof", "traffic", "traffic_source", "traffic_build", "traffic_stale", "algorithmic_bytes_per_launch"):
            assert k in g, (leg, k)
    assert c["c3"]["algorithmic_bytes_per_launch"] == 4096 * 8192 * 8 and c["c4_share"]["algorithmic_bytes_per_launch"] == 2048 * 4096 * 12
    assert c["c4_share"]["submits_in_flight"] == 3
    # C5's timed output is held to the oracle in the same run ...
    assert c["c5"]["c5_check"]["match"] is True and c["c5"]["c5_check"]["psds_checked"] == [0, 31]
    # ... and the C4 share runs once more in steady state with the ONE collective inside the timed region
    check_gather_leg(c["c4_share_gather"], centres=2048, per_gpu=2048)


def check_gather_leg(g, centres, per_gpu):
    for k in ("sweep_us", "sweep_with_gather_us", "exposed_gather_us", "gather_us", "records_per_sweep", "lists_gathered", "cap_per_rank"):
        assert k in g and isinstance(g[k], (int, float)), k
    assert g["scaling"] == "strong" and g["centres"] == centres and g["centres_per_gpu"] == per_gpu and g["n_gpus"] == 1
    assert g["sweep_us"] > 0 and g["gather_us"] > 0 and abs(g["exposed_gather_us"] - (g["sweep_with_gather_us"] - g["sweep_us"])) < 0.02
    S = g["sweeps_per_launch"]                       # a shard below 8192 buffers is launched S sweeps at a time: the last LAUNCH's list is checked
    assert S == max(1, 8192 // per_gpu) and g["buffers_per_launch"] == min(per_gpu, 8192) * S
    assert g["check"]["match"] is True and g["check"]["gathered_hits"] == g["check"]["expected_hits"] == 7 * (centres // 4) * S
    assert g["records_per_sweep"] == 7 * (centres // 4) and g["lists_gathered"] == round(g["steps"] * g["launches_per_sweep"])
    assert "scn_gather_post" in g["transport"]


def test_c4_line_with_a_gather_every_sweep():
    """`bench.py --config c4 --gather-every-sweep`: the strong-scaling sweep with every launch's hit list gathered to rank 0 inside
    the timed region, beside the same sweeps without the gather (one rank here: the communicator has no peers)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "c4", "--centres", "4096", "--steps", "4", "--warmup", "1",
                          "--settle", "0.05", "--no-cpu-baseline", "--no-overlap-leg", "--no-records-leg", "--no-hits-only-leg", "--no-copy-ref",
                          "--gather-every-sweep"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads(out.stdout.strip())
    assert d["scaling"] == "strong" and d["c4_check"]["match"] is True
    check_gather_leg(d["gather_every_sweep"], centres=4096, per_gpu=4096)


def test_bench_on_the_workers_kind_of_plan_and_in_time_domain_mode():
    """what scripts/prof_all.sh profiles for the product's own worker: a hits-only plan (with DC removal), the time-domain kernel"""
    d = run_bench(["--n", "8192", "--kind", "int16", "--plan-mode", "hits", "--dc", "--no-cpu-baseline"])
    check_common(d)
    assert d["roofline"]["kernel"] == "scn_fft8k_kernel<SCN_K_SHORT_COMPLEX, true, true, false>"
    assert d["roofline"]["algorithmic_bytes_per_sample"] == 4 and d["config"]["plan_flags"] == "SCN_OUT_HITS"
    assert d["config"]["plan_mode"] == "hits" and d["config"]["correct_dc"] is True and d["final_sweep_hits"] > 0
    assert d["hits_only"] is None and d["overlap"] is None and d["with_hit_records"] is None
    d = run_bench(["--n", "8192", "--kind", "int8", "--time-domain", "--no-cpu-baseline"])
    check_common(d)
    assert d["roofline"]["kernel"] == "scn_time_domain_wave_kernel<SCN_K_BYTE_COMPLEX, false>"
    assert d["roofline"]["algorithmic_bytes_per_sample"] == 2 and d["config"]["time_domain"] is True


def test_bench_with_the_centres_per_buffer_and_from_the_table():
    """the default legs name a run of the plan's GPU-resident frequency table per submit (scn_submit_device_indexed);
    --per-buffer-centres sends them with every submit as until round 4: the same detections either way"""
    extra = ["--n", "128", "--no-cpu-baseline", "--no-configs-leg"]
    t = run_bench(extra, batch=("--batch", "4096"))
    b = run_bench(extra + ["--per-buffer-centres"], batch=("--batch", "4096"))
    check_common(t)
    check_common(b)
    assert "table" in t["config"]["centre_frequencies"] and "per buffer" in b["config"]["centre_frequencies"]
    assert t["final_sweep_hits"] == b["final_sweep_hits"] > 0
    # (the records legs driven from the bench's own process go through the same two forms)
    assert t["with_hit_records"]["python_torch_runtime"]["hits_per_step"] == b["with_hit_records"]["python_torch_runtime"]["hits_per_step"] > 0


@pytest.mark.parametrize("extra,kind", [([], "cfloat"), (["--kind", "int16", "--dc"], "int16"), (["--kind", "int8", "--welch-pinned"], "int8")])
def test_welch_line_checks_its_own_output(extra, kind):
    """`bench.py --welch` (BASELINE C5): the first and last PSD of the last TIMED step are held to the oracle outside the timed
    region (a mismatch exits non-zero and prints no line), for float samples and for the integer wire formats, device-resident and
    through pinned memory + the captured hipGraph."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--welch", "--welch-psd", "8", "--steps", "4", "--warmup", "1"] + extra,
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads(out.stdout.strip())
    for k, t in CONTRACT.items():
        assert k in d and isinstance(d[k], t), k
    assert d["config"]["kind"] == kind and d["config"]["psd_per_submit"] == 8 and d["config"]["workload"].startswith("C5:")
    c = d["c5_check"]
    assert c["match"] is True and c["psds_checked"] == [0, 7] and c["max_rel_power_vs_max_bin_mean"] <= c["bar"] == 1e-5
    r = d["roofline"]
    want = 8 * 16 * 32768 * {"cfloat": 8, "int16": 4, "int8": 2}[kind] + 8 * 65536 * 4     # bytes of NEW samples + the PSDs
    assert r["bound"] == "hbm" and r["algorithmic_bytes_per_launch"] == want and 0 < r["frac"] < 1

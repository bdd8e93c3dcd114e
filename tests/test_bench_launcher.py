"""`python bench.py --gpus N` launches itself (a parent that never touches the GPU spawns one rank per GPU through
torch.distributed.run, forwards the single JSON line and exits non-zero if any rank fails).  Exercised here on CPU with
--dry-run: the ranks use gloo, shard the C4 table, and gather the planted-emitter hit lists; plus the closed-form C4
expectation itself against the oracle on a small sweep."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run(args, timeout=300):
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, timeout=timeout, cwd=ROOT)


def test_self_launch_two_ranks_dry_run(built_lib):
    out = run(["--gpus", "2", "--dry-run", "--config", "c4", "--centres", "300"])
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["dry_run"] is True and d["scaling"] == "strong"
    c = d["c4_check"]
    assert c["match"] is True and c["expected_hits"] == c["gathered_hits"] == 75 * 7
    assert sum(c["per_rank"]) == c["gathered_hits"] and len(c["per_rank"]) == 2 and min(c["per_rank"]) > 0
    g = d["gather_every_sweep"]       # the steady-state form: a list per sweep, two posts in flight
    assert g["match"] is True and g["lists_gathered"] == g["sweeps"] == 7


def test_self_launch_three_ranks_ragged(built_lib):
    d = json.loads(run(["--gpus", "3", "--dry-run", "--config", "c4", "--centres", "1000"]).stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 3 and d["c4_check"]["match"] is True and len(d["c4_check"]["per_rank"]) == 3


def test_self_launch_reports_rank_failure(built_lib):
    # a threshold on top of a leakage bin makes every rank raise: the parent must not print a result line
    out = run(["--gpus", "2", "--dry-run", "--config", "c4", "--centres", "64", "--threshold", "10.7"])
    assert out.returncode != 0
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_single_process_dry_run_needs_no_launcher(built_lib):
    out = run(["--dry-run", "--config", "c4", "--centres", "128"])
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads(out.stdout.strip())
    assert d["n_gpus"] == 1 and d["c4_check"]["match"] is True


def test_c4_closed_form_matches_oracle(oracle_mod, built_lib):
    """The expectation the C4 GPU runs are held to is itself checked against the oracle: a 96-centre sweep with the
    planted emitters, run through the CPU restatement of process.cpp, gives exactly that list."""
    from scanner_amd import capi, synth

    n, fs, n_centres, thr = 4096, 8000000, 96, 10.0
    _, fc = capi.frequency_table(fs, 0.0, n_centres * 0.75 * fs)
    centres, i0 = synth.c4_emitters(n_centres, n)
    x = synth.c4_shard(n, 0, n_centres, centres, i0, seed=4)
    _, hits, trig = oracle_mod.Oracle(n, fs, thr).run(x, fc, np.arange(n_centres, dtype=np.uint64))
    want = synth.c4_expected_hits(synth.blackman_harris(n), fc, centres, i0, n, fs, thr)
    assert len(want) == 7 * len(centres) == len(hits)
    for f in ("seq_id", "i", "freq_hz"):
        assert np.array_equal(hits[f], want[f]), f
    assert np.abs(hits["power_db"] - want["power_db"]).max() < 0.15
    assert not trig.any()
    # the lowest bins of a sweep that starts at 0 Hz have NEGATIVE frequencies (start = 3 MHz - 4 MHz); the cast to
    # uint64 (process.cpp:57) then follows x86-64: pin the helper that the GPU compaction kernel mirrors
    e = synth.c4_expected_hits(synth.blackman_harris(n), fc, np.array([0]), np.array([515]), n, fs, thr)
    assert e["freq_hz"][0] == np.uint64(np.int64(-1e6 + 512 * 1953)) and e["freq_hz"][-1] < 1 << 40


def test_self_launch_eight_ranks_both_configs_one_shard_empty(built_lib):
    """What the driver's N = 8 run does, minus the kernels: `bench.py --gpus 8` launches itself, every rank takes its shard --
    16384 / 8 centres of the C4 table (strong scaling), or its own batch of the C2-shaped table (weak scaling) --, rank 5's
    shard is quiet, and rank 0 receives the rank-major concatenation of the other seven lists."""
    for cfg, extra, scaling, centres in (("c4", ["--centres", "2048"], "strong", 2048), ("c2", ["--batch", "64"], "weak", 8 * 64)):
        out = run(["--gpus", "8", "--dry-run", "--config", cfg, "--dry-run-quiet-rank", "5"] + extra, timeout=600)
        assert out.returncode == 0, out.stderr[-3000:]
        lines = [l for l in out.stdout.splitlines() if l.strip()]
        assert len(lines) == 1, out.stdout
        d = json.loads(lines[0])
        assert d["n_gpus"] == 8 and d["scaling"] == scaling and d["config"]["centres"] == centres
        c = d["c4_check"]
        assert c["match"] is True and len(c["per_rank"]) == 8 and c["per_rank"][5] == 0
        assert all(v > 0 for r, v in enumerate(c["per_rank"]) if r != 5) and sum(c["per_rank"]) == c["gathered_hits"] == c["expected_hits"]
        g = d["gather_every_sweep"]   # ... and so does every sweep's list in the steady-state form (post / wait, two in flight)
        assert g["match"] is True and g["lists_gathered"] == g["sweeps"] == 7

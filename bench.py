#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X spectrum-scan path.

Metric (BASELINE.json): Msamples/s of complex samples through the whole per-buffer path
(convert -> window -> 4096-pt FFT -> dB -> per-bin threshold), plus swept GHz/s and the
achieved fraction of the HBM roofline.  Workload at N=1 is BASELINE config C2:
4096-pt FFT, batch 8192 synthetic cfloat buffers resident in HBM, one MI355X.

A "step" is one pass of the hot path over one batch = ONE kernel launch through the C-ABI
(scn_submit_device), double-buffered over the plan's two slots; the per-buffer hit counts /
trigger flags are collected every step (scn_collect), the dB spectra stay in HBM.
Steps rotate over R distinct input batches and R output buffers (>= 1.5 GiB in total), so no
step can find its input or leave its output in the 256 MiB Infinity Cache: with a single
re-read batch the same kernel looks 15 % faster than HBM can actually feed it.  With --gpus N (one process per GPU, torch.distributed / RCCL) every rank owns a
contiguous range of the frequency table (its own batch: weak scaling, no data-path
collective); the final hit list is gathered to rank 0 once, after the timed region.

  python bench.py                                   # 1 GPU
  python bench.py --gpus 8                          # launches itself: a parent that never touches the GPU spawns
                                                    # `python -m torch.distributed.run ... bench.py --gpus 8` and forwards
                                                    # the one JSON line (non-zero exit if any rank fails)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
         --master-port 29500 bench.py --gpus 8      # what the parent runs (and what the driver may run itself)
  python bench.py --config c4 --gpus 8              # BASELINE config 4: 16384 centres x 4096-pt, 16384/N per rank
                                                    # (strong scaling), planted emitters, gathered list checked on rank 0
  python bench.py --gpus 2 --dry-run                # CPU/gloo run of the launcher, sharding, gather and JSON plumbing
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
FS = 8000000           # scan.cpp:92 default sample rate
USE_BW = 0.75          # scan.cpp:65


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--settle", type=float, default=0.4,
                    help="seconds of untimed launches BEFORE the warmup steps, so that the GPU has left its idle power "
                         "state whatever --warmup is (from idle the first ~100 launches run at half speed and a ~35 ms "
                         "power-management stall follows around launch 700-800, scripts/drift.py); 0 disables")
    ap.add_argument("--n", type=int, default=4096)
    ap.add_argument("--batch", type=int, default=8192)
    ap.add_argument("--kind", default="cfloat", choices=["cfloat", "int16", "int16p", "int8"],
                    help="wire format (int16p: planar int16, the SDRplay layout of utility.cpp:9-32)")
    ap.add_argument("--plan-mode", default="both", choices=["both", "hits", "spectrum"],
                    help="what the main leg's plan reports: both (SCN_OUT_SPECTRUM|SCN_OUT_HITS, the default and the only mode `value` "
                         "of the driver's line is quoted on), hits (SCN_OUT_HITS alone: what ProcessSamples::ThreadWorker creates -- no "
                         "spectrum is stored, the algorithmic bytes are the raw samples alone), spectrum (SCN_OUT_SPECTRUM alone)")
    ap.add_argument("--dc", action="store_true", help="integer formats: remove the integer mean first (utility.cpp:70-79, correctDC)")
    ap.add_argument("--time-domain", action="store_true",
                    help="the reference CLI's default mode (scan.cpp:87, process.cpp:203-237): per-buffer max / min dB, no FFT")
    ap.add_argument("--per-buffer-centres", action="store_true",
                    help="send the centre frequencies with every submit (scn_submit_device) instead of naming a run of the plan's "
                         "GPU-resident frequency table (scn_plan_set_table + scn_submit_device_indexed)")
    ap.add_argument("--no-configs-leg", action="store_true",
                    help="skip the short legs for BASELINE configs C3, the C4 per-GPU share and C5 that ride on the default C2 line (`configs`)")
    ap.add_argument("--threshold", type=float, default=None,
                    help="dB; default: 10 dB at 4096 points and the same margin over the noise mean at every other size "
                         "(10 + 5 log10(n / 4096): a bin's noise power grows with n, and a fixed 10 dB sits UNDER the noise mean of a "
                         "65536-point buffer -- 30 %% of the bins became hits and the large sizes' numbers measured hit recording)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=6.0,
                    help="approximate wall budget of the CPU baseline (three legs: 1, 2 and 8 threads, ~20 CPU-seconds)")
    ap.add_argument("--rotate", type=int, default=0, help="distinct input/output batches (0: enough for 1.5 GiB)")
    ap.add_argument("--no-overlap-leg", action="store_true",
                    help="skip the extra leg that times the same steps on a plan with SCN_PLAN_OVERLAP_SLOTS "
                         "(reported separately under \"overlap\"; value / roofline are always the single-stream plan)")
    ap.add_argument("--config", default="c2", choices=["c2", "c4"],
                    help="c2 (default): --batch buffers per GPU, weak scaling.  c4: BASELINE config 4, the full frequency table "
                         "FrequencyTable(8e6, 0, 16384*6e6) sharded 16384/N per rank (strong scaling), one 4096-pt cfloat buffer "
                         "per centre, emitters planted at known absolute frequencies, a step = one sweep of the rank's shard, "
                         "the gathered hit list is compared with the computed expectation on rank 0")
    ap.add_argument("--centres", type=int, default=16384, help="c4: number of centre frequencies in the table")
    ap.add_argument("--sweeps-per-launch", type=int, default=0,
                    help="c4: sweeps of the rank's shard batched into one launch (0: as many as fit 8192 buffers; 1: a launch per sweep)")
    ap.add_argument("--gather-every-sweep", action="store_true",
                    help="c4: add the steady-state leg -- a launch per sweep, every launch's ordered hit list gathered to rank 0 inside the timed "
                         "region (scn_gather_post / scn_gather_wait on a communicator created once), beside the same sweeps without the gather "
                         "(the default --gpus N > 1 line carries this leg as configs.c4)")
    ap.add_argument("--no-hits-only-leg", action="store_true", help="skip the extra leg on a plan without SCN_OUT_SPECTRUM")
    ap.add_argument("--no-copy-ref", action="store_true", help="skip the device-to-device copy measured beside the roofline")
    ap.add_argument("--records-depth", type=int, default=4, help="records legs: submits in flight (<= SCN_NUM_SLOTS)")
    ap.add_argument("--no-records-leg", action="store_true",
                    help="skip the extra leg that times the same steps with the ordered hit records fetched every step")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU: every rank (gloo) shards the table, fabricates its shard's expected hit list, the list is "
                         "gathered and checked on rank 0 -- exercises the launcher, rendezvous, sharding, gather and the JSON line")
    ap.add_argument("--dry-run-quiet-rank", type=int, default=-1,
                    help="dry run: this rank's shard reports no detection (a quiet part of the band): its empty list still takes part in the gather")
    ap.add_argument("--master-port", type=int, default=0, help="self-launch: rendezvous port (0: pick a free one)")
    ap.add_argument("--welch", action="store_true", help="BASELINE config C5: streaming 65536-pt 50%%-overlap Welch PSD")
    ap.add_argument("--welch-psd", type=int, default=32, help="PSDs per submit (K=16 segments each)")
    ap.add_argument("--welch-pinned", action="store_true", help="feed from pinned host memory through the captured hipGraph")
    args = ap.parse_args()
    if args.threshold is None:
        args.threshold = round(10.0 + 5.0 * np.log10((4096 if args.config == "c4" else args.n) / 4096.0), 2)
    return args


def cpu_baseline(args, raw_host, kind_oracle, enob, budget_s, dc=False):
    """Times the oracle (CPU restatement of process.cpp/fft.cpp/utility.cpp, its own FFT --
    NOT FFTW) on a bounded sample of the same buffers.  Checker code used as the reported
    baseline only; never on the product path."""
    from oracle import oracle as O

    O.build()
    O.set_fft_mode(False)  # the float radix-2 FFT: FFTW computes in float too; the double-internal mode is for parity
    n = args.n
    o = O.Oracle(n, FS, args.threshold, kind=kind_oracle, enob=enob, correct_dc=dc)
    ncores = os.cpu_count() or 1
    tmax = min(8, ncores)  # the reference's cap, process.h:49
    res = {}
    # calibrate on a few buffers, then size each leg to ~budget/3
    t0 = time.perf_counter()
    o.run(raw_host[:32], want_power=True, want_hits=True, threads=1)
    per_buf = (time.perf_counter() - t0) / 32
    cpu_s = 0.0
    for t in sorted({1, 2, tmax}):
        nb = int(min(len(raw_host), max(64, (budget_s / 3) / per_buf * t)))
        reps = max(1, int(round((budget_s / 3) / (per_buf * nb / t))))  # passes over the sample: ~budget/3 of wall per leg
        t0 = time.perf_counter()
        for _ in range(reps):
            o.run(raw_host[:nb], want_power=True, want_hits=True, threads=t)
        dt = time.perf_counter() - t0
        cpu_s += dt * t
        res[t] = (reps * nb * n / dt / 1e6, nb, reps)
    v, nb, reps = res[tmax]
    return {
        "value": round(v, 3), "unit": "Msamples/s", "cores": tmax, "kind": "port",
        "sample": f"{reps} passes over the first {nb} of the {args.batch} buffers of rank 0's batch, {n}-pt, "
                  f"oracle/scn_oracle.c (own radix-2 FFT, not FFTW), spectra+hits, stdout suppressed; "
                  f"~{cpu_s:.0f} CPU-seconds over the 1/2/{tmax}-thread legs",
        "host_cores_available": ncores,
        "threads_1": round(res[1][0], 3), "threads_2": round(res[min(2, tmax)][0], 3),
    }


class ParityError(Exception):
    """A timed leg's output differs from the oracle's: the number is void and the run exits non-zero."""


WELCH_KINDS = {"cfloat": (4, 8, None), "int16": (3, 4, 2047.0), "int16p": (2, 4, 2047.0), "int8": (1, 2, 127.0)}   # SCN_KIND, B/sample, full scale


def welch_leg(torch, dev, local_rank, seed, npsd, steps, warmup, rotate=0, pinned=False, sync=None, kind="cfloat", dc=False, check=None):
    """C5 steps on this GPU: a step = one submit of `npsd` PSDs = (npsd*16 + 1) * 32768 complex samples, device-resident (rotated
    over R streams past the Infinity Cache) or, with `pinned`, staged from pinned host memory through the captured hipGraph
    (PCIe-bound).  Returns (seconds for `steps` steps, new samples per step); sync() brackets the timed region (N > 1).
    check: a dict to receive the comparison of the LAST timed step's first and last PSD with the CPU oracle (outside the timed
    region; raises ParityError when they differ by more than the parity bar)."""
    from scanner_amd import WelchPlan

    N, K = 65536, 16
    okind, bps, full_scale = WELCH_KINDS[kind]
    enob = 8 if kind == "int8" else 12
    plan = WelchPlan(N, K, max_psd=npsd, device_id=local_rank, kind=okind, enob=enob, correct_dc=dc)
    m = plan.samples(npsd)
    new_samples = npsd * K * (N // 2)
    R = rotate or max(2, -(-(3 << 29) // (m * bps)))
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    blk = torch.rand((m // (N // 2), 1, 1), generator=g, device=dev) * 0.6 + 0.4   # a level per delivery block: a PSD built from the wrong segments shows

    def stream():
        x = (torch.randn((m // (N // 2), N // 2, 2), generator=g, device=dev) * 0.05) * blk
        if full_scale is None:
            return x.reshape(m, 2).contiguous()
        dt = torch.int8 if kind == "int8" else torch.int16
        q = torch.clamp(torch.round(x * full_scale), -full_scale - 1, full_scale).to(dt)
        if kind == "int16p":   # planar per delivery block: I[hop] then Q[hop]
            q = q.permute(0, 2, 1)
        return q.contiguous().reshape(-1)

    if pinned:
        host_in = []
        for s in range(2):
            hb = plan.host_buffer(s)
            flat = stream().cpu().numpy().view(np.uint8).reshape(-1)
            hb.view(np.uint8)[: flat.size] = flat
            host_in.append(flat)
    else:
        xs = [stream() for _ in range(R)]
        outs = [torch.empty((npsd, N), dtype=torch.float32, device=dev) for _ in range(R)]
    torch.cuda.synchronize()
    pending = [False, False]

    def step(k):
        s = k & 1
        if pending[s]:
            plan.collect(s, want_psd=False)
        if pinned:
            plan.submit(s, npsd)
        else:
            plan.submit_device(s, xs[k % R], npsd, d_psd_db=outs[k % R], sync_producer=False)
        pending[s] = True

    def drain():
        for s in (0, 1):
            if pending[s]:
                plan.collect(s, want_psd=False)
                pending[s] = False

    for k in range(warmup):
        step(k)
    drain()
    if sync:
        sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        step(k)
    drain()
    torch.cuda.synchronize()
    if sync:
        sync()
    elapsed = time.perf_counter() - t0
    if check is not None and steps > 0:
        # the LAST timed step's first and last PSD against the CPU oracle (K1 per delivery block, window, FFT, mean |X|^2, dB)
        from oracle import oracle as O
        from tests import tolerances as tol

        last = steps - 1
        if pinned:
            plan.submit(last & 1, npsd)       # the slot's pinned input is still the last step's: replay it for its PSDs
            got = plan.collect(last & 1)
            raw = host_in[last & 1]
        else:
            got = outs[last % R].cpu().numpy()
            raw = xs[last % R].cpu().numpy().view(np.uint8).reshape(-1)
        hop_b = (N // 2) * bps
        worst = 0.0
        try:
            for p in sorted({0, npsd - 1}):
                ref = O.welch_raw(raw[p * K * hop_b:(p * K + K + 1) * hop_b], okind, enob, dc, N, K, 1)
                worst = max(worst, tol.compare_spectra(got[p:p + 1], ref)["max_rel_power_vs_max_bin_mean"])
            check.update({"match": True, "psds_checked": sorted({0, npsd - 1}), "max_rel_power_vs_max_bin_mean": worst, "bar": tol.REL_POWER,
                          "against": "oracle (CPU restatement), the last timed step's output"})
        except AssertionError as e:
            check.update({"match": False, "detail": str(e)[:300]})
            plan.close()
            raise ParityError(f"C5 output differs from the oracle: {str(e)[:300]}")
    plan.close()
    return elapsed, new_samples


def c4_gather_leg(torch, dev, local_rank, rank, world, n_centres, steps, warm=60, sync=None, threshold=10.0, sweeps_per_launch=0):
    """BASELINE config 4 in its steady state, the ONE collective inside the timed region: this rank sweeps its contiguous shard of the
    n_centres-entry table once per step (launches of at most 8192 buffers; a shard smaller than that is launched S sweeps at a time --
    a launch is a batch of whatever is queued, as in the main C4 loop -- unless sweeps_per_launch says otherwise; four slots in
    flight, each on its own stream below 8192 buffers), and every launch's ordered hit list goes to rank 0 through scn_gather_post / scn_gather_wait on
    a communicator created ONCE, before the region -- posted two launches behind the newest submit, waited for two launches later,
    so that the exchange runs beside the next sweeps' kernels.  Timed twice over the same launches: without the gather (the
    counts are collected at the same place) and with it.  Returns the figures; raises ParityError when the root's last list
    differs from the planted emitters' closed form."""
    import ctypes as C

    from scanner_amd import Plan, capi, sweep, synth

    n = 4096
    _, fc_all = capi.frequency_table(FS, 0.0, n_centres * USE_BW * FS, USE_BW, 0.0)
    first, fc = capi.frequency_table(FS, 0.0, n_centres * USE_BW * FS, USE_BW, 0.0, shard=rank, n_shards=world)
    shard = len(fc)
    # what every rank must agree on -- the number of posts per sweep and the size of a message -- comes from the LARGEST shard (ranks of
    # a ragged split differ by one centre; N = 1, 2, 4, 8 split 16384 evenly)
    max_shard = -(-n_centres // world)
    S = (max(1, 8192 // max_shard) if sweeps_per_launch == 0 else sweeps_per_launch) if max_shard < 8192 else 1
    nb = min(shard, 8192) * S                # buffers per launch
    chunks = [(lo, min(lo + 8192, shard)) for lo in range(0, shard, 8192)] if S == 1 else [(0, nb)]
    n_chunks_all = -(-max_shard // 8192) if S == 1 else 1
    assert len(chunks) == n_chunks_all, "ranks would post different numbers of lists per sweep"
    centres, i0 = synth.c4_emitters(n_centres, n)
    step_bytes = shard * n * 12
    R = max(2, -(-(3 << 29) // step_bytes))
    R = -(-R // S) * S
    raws = [synth.c4_shard_torch(n, first, shard, centres, i0, seed=4 + 1000 * r, device=dev) for r in range(R)]
    if S > 1:   # S consecutive sweeps back to back in memory = one launch; ids run on from sweep to sweep (messageQueue.h:86)
        raws = [torch.cat(raws[g * S:(g + 1) * S], dim=0).contiguous() for g in range(R // S)]
        R = R // S
    outs = [torch.empty((nb if S > 1 else shard, n), dtype=torch.float32, device=dev) for _ in range(R)]
    seq = np.concatenate([np.arange(first, first + shard, dtype=np.uint64) + np.uint64(j * n_centres) for j in range(S)])
    steps = -(-steps // S) * S * (2 if S > 1 else 1)   # whole launches, and no fewer than a hundred of them
    warm = -(-warm // S) * S
    D, LAG = 4, 2
    hit_cap = nb * 64
    cap = 7 * (-(-(min(max_shard, 8192) * S) // 4) + S) * 2   # per launch: an emitter on every 4th centre, seven bins each, and as much again (the same on every rank)
    flags = capi.OUT_SPECTRUM | capi.OUT_HITS | (capi.PLAN_OVERLAP_SLOTS if nb < 8192 else 0)
    plan = Plan(n, FS, threshold, max_batch=min(nb, 8192 * S), max_hits=hit_cap, device_id=local_rank, flags=flags)
    plan.set_table(fc)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    prep = [[(C.c_void_p(raws[r][lo:hi].data_ptr()), hi - lo, lo, vp(seq[lo:hi]), C.c_void_p(outs[r][lo:hi].data_ptr())) for lo, hi in chunks]
            for r in range(R)]
    g = sweep.HitGather(dev)               # rendezvous + ncclCommInitRank: once, outside every timed region
    L = capi.lib()
    post, wait_ = L.scn_gather_post, L.scn_gather_wait
    comm, ph = g._comm, plan.handle
    tk = C.c_uint32()
    lst, tot = C.c_void_p(), C.c_uint64()
    per_rank = np.zeros(world, np.uint32)
    per_rank_p = vp(per_rank)
    state = {"launch": 0, "base": 0, "records": 0, "lists": 0, "wait_s": 0.0, "post_s": 0.0, "collect_s": 0.0, "last": None, "keep": False}
    tickets = [None] * D
    pending = [False] * D

    def finish(s):                         # the gather of the launch in slot s has arrived (root: the list is read where it lies)
        t0 = time.perf_counter()
        st = wait_(comm, tickets[s], C.byref(lst), C.byref(tot), per_rank_p)
        state["wait_s"] += time.perf_counter() - t0
        if st:
            capi.check(st, "scn_gather_wait")
        tickets[s] = None
        state["records"] += tot.value
        state["lists"] += 1
        if state["keep"] and rank == 0:
            raw = np.ctypeslib.as_array(C.cast(lst, C.POINTER(C.c_uint8)), shape=(tot.value * 24,)).view(capi.HIT_DTYPE) if tot.value else np.zeros(0, capi.HIT_DTYPE)
            state["last"].append(raw.copy())

    def launch(k, a, gather):
        j = state["launch"]
        s = j % D
        if tickets[s] is not None:
            finish(s)
        elif pending[s]:
            plan.collect_counts(s)
            pending[s] = False
        plan.submit_prepared_indexed(s, *a)
        pending[s] = True
        state["launch"] = j + 1
        if j - LAG >= state["base"]:           # (launches before `base` were drained)
            s2 = (j - LAG) % D
            tc0 = time.perf_counter()
            plan.collect_counts(s2)
            tc1 = time.perf_counter()
            state["collect_s"] += tc1 - tc0
            pending[s2] = False
            if gather:
                st = post(comm, ph, s2, 0, cap, C.byref(tk))
                state["post_s"] += time.perf_counter() - tc1
                if st:
                    capi.check(st, "scn_gather_post")
                tickets[s2] = tk.value

    def drain(gather):
        j = state["launch"]
        for jj in range(max(j - LAG, state["base"]), j):   # the launches not yet collected, oldest first
            s2 = jj % D
            if pending[s2]:
                plan.collect_counts(s2)
                pending[s2] = False
                if gather:
                    st = post(comm, ph, s2, 0, cap, C.byref(tk))
                    if st:
                        capi.check(st, "scn_gather_post")
                    tickets[s2] = tk.value
        for jj in range(max(j - D, 0), j):
            if tickets[jj % D] is not None:
                finish(jj % D)
        state["base"] = j

    def run(k_steps, gather, keep_last=False):
        k_steps //= S                          # launches of S sweeps
        for k in range(k_steps):
            if keep_last and k == k_steps - 1:
                drain(gather)                  # everything before the last sweep has arrived; now record the last sweep's lists
                state["keep"], state["last"] = True, []
            for a in prep[k % R]:
                launch(k, a, gather)
        drain(gather)
        state["keep"] = False

    def timed(gather):
        run(warm, gather)
        torch.cuda.synchronize()
        if sync:
            sync()
        state["records"] = state["lists"] = 0
        state["wait_s"] = state["post_s"] = state["collect_s"] = 0.0
        t0 = time.perf_counter()
        run(steps, gather, keep_last=gather)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        state["host_" + ("gather" if gather else "plain")] = {k: round(state[k] / steps * 1e6, 2) for k in ("wait_s", "post_s", "collect_s")}
        if sync:
            sync()
        if world > 1:
            import torch.distributed as dist
            tt = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = tt.item()
        return el

    try:
        el_plain = timed(False)
        el_gather = timed(True)
        lists, records, wait_s = state["lists"], state["records"], state["wait_s"]
        last = np.concatenate(state["last"]) if (rank == 0 and state["last"]) else np.zeros(0, capi.HIT_DTYPE)
        # the gather alone, nothing to hide behind: post + wait back to back on an idle GPU (every rank takes part)
        plan.submit_prepared_indexed(0, *prep[0][0])
        plan.collect_counts(0)
        lat = []
        for _ in range(40):
            t0 = time.perf_counter()
            capi.check(post(comm, ph, 0, 0, cap, C.byref(tk)), "scn_gather_post")
            capi.check(wait_(comm, tk.value, C.byref(lst), C.byref(tot), per_rank_p), "scn_gather_wait")
            lat.append((time.perf_counter() - t0) * 1e6)
        out = None
        if rank == 0:
            want = synth.c4_expected_hits(plan.window(), fc_all, centres, i0, n, FS, threshold)
            if S > 1:   # the last launch holds S sweeps: the same records, the ids running on
                parts = []
                for j in range(S):
                    w = want.copy()
                    w["seq_id"] += np.uint64(j * n_centres)
                    parts.append(w)
                want = np.concatenate(parts)
            ok, worst = compare_hit_lists(last, want)
            sweep_us = el_plain / steps * 1e6
            gsweep_us = el_gather / steps * 1e6
            out = {"value": round(n_centres * n * steps / el_gather / 1e6, 1), "unit": "Msamples/s", "steps": steps, "scaling": "strong",
                   "centres": n_centres, "centres_per_gpu": shard, "launches_per_sweep": len(chunks) if S == 1 else round(1.0 / S, 4), "sweeps_per_launch": S,
                   "buffers_per_launch": chunks[0][1] - chunks[0][0], "n_gpus": world,
                   "sweep_us": round(sweep_us, 2), "sweep_with_gather_us": round(gsweep_us, 2),
                   "exposed_gather_us": round(gsweep_us - sweep_us, 2), "exposed_frac": round((gsweep_us - sweep_us) / sweep_us, 4),
                   "gather_us": round(float(np.median(lat)), 1), "gather_us_min": round(float(min(lat)), 1),
                   "value_without_gather": round(n_centres * n * steps / el_plain / 1e6, 1),
                   "lists_gathered": lists, "records_per_sweep": round(records / max(steps, 1), 1), "cap_per_rank": cap, "lists_per_sweep": round(lists / max(steps, 1), 4),
                   "root_blocked_in_wait_us_per_sweep": round(wait_s / steps * 1e6, 2),
                   "host_us_per_sweep": {"without_gather": state.get("host_plain"), "with_gather": state.get("host_gather"),
                                         "note": "this rank's host time per sweep inside scn_collect (waiting for the launch two behind), scn_gather_post, scn_gather_wait"},
                   "check": {"expected_hits": int(len(want)), "gathered_hits": int(len(last)), "match": bool(ok), "max_power_db_diff": worst,
                             "of": "the LAST timed sweep's gathered list against the planted emitters' closed form"},
                   "transport": ("scn_gather_post / scn_gather_wait: one fixed-size message per rank per launch (header + cap records), ONE group of "
                                 "ncclSend / ncclRecv on the communicator's stream, compaction into the root's pinned list; posted two launches "
                                 "behind the newest submit, waited for two launches later; communicator created once before the region"
                                 + ("; one rank: no peers, the root's own part only" if world == 1 else "")),
                   "workload": f"C4: {n_centres} centres x 4096-pt cfloat, {shard} per GPU per sweep over {world} GPU(s), a launch per "
                               f"{('sweep' if S == 1 else str(S) + ' sweeps') if len(chunks) == 1 else 'half sweep'}, four slots in flight"
                               + (", each on its own stream (SCN_PLAN_OVERLAP_SLOTS)" if nb < 8192 else "") + ", the hit list of every launch gathered to rank 0"}
            if not ok:
                raise ParityError(f"C4 gathered list differs from the closed form: {out['check']}")
        return out
    finally:
        g.close()
        plan.close()


def welch_roofline(elapsed, steps, new_samples, npsd, kind="cfloat", dc=False):
    N, K = 65536, 16
    algo = new_samples * WELCH_KINDS[kind][1] + npsd * N * 4  # 8 (4, 2) B per NEW sample + 4N/K per segment
    ms = elapsed / steps * 1e3
    achieved = algo / (ms * 1e-3) / 1e9
    prof = _tracked(f"welch/{N}/{K}/{npsd}" + ("" if kind == "cfloat" else f"/{kind}") + ("/dc" if dc and kind != "cfloat" else ""))
    r = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": round(achieved / HBM_PEAK_GBS, 4), "frac_is": "frac_wall (host wall time per step: the Welch plan exposes no stream to put events on)",
         "kernel": "scn_welch_cols_kernel + scn_welch_rows_kernel (two-pass four-step FFT: the work "
                   "buffer round trip and the 50% overlap re-read are NOT algorithmic bytes)",
         "kernels_avg_us_rocprof": prof.get("kernels"), "algorithmic_bytes_per_launch": algo}
    r.update(roofline_from_profile(prof, algo))
    return r, ms


def welch_main(args):
    """C5: every rank processes its own stream (replicas, no collective)."""
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    npsd = args.welch_psd
    check = {} if rank == 0 else None
    rc = 0
    try:
        elapsed, new_samples = welch_leg(torch, dev, local_rank, 5 + rank, npsd, args.steps, args.warmup, args.rotate, args.welch_pinned,
                                         sync=dist.barrier if world > 1 else None, kind=args.kind, dc=args.dc, check=check)
    except ParityError as e:
        print(f"bench.py: {e}", file=sys.stderr)
        sys.exit(3)
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = tt.item()
    if rank == 0:
        roof, ms = welch_roofline(elapsed, args.steps, new_samples, npsd, args.kind, args.dc)
        emit({
            "metric": "Msamples/s (new complex samples through 65536-pt 50%-overlap Welch PSD)",
            "value": round(world * new_samples * args.steps / elapsed / 1e6, 1), "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 5),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"C5: 65536-pt 50%-overlap Welch PSD, K=16, {npsd} PSDs per submit, Blackman-Harris, {args.kind} samples"
                                   f"{' with DC removal per delivery block' if args.dc and args.kind != 'cfloat' else ''}, "
                                   f"{'pinned host staging + hipGraph replay' if args.welch_pinned else 'stream resident in HBM'}; "
                                   f"independent stream per GPU (replicas)",
                       "n": 65536, "segments_per_psd": 16, "psd_per_submit": npsd, "pinned": bool(args.welch_pinned), "kind": args.kind,
                       "sample_kind": args.kind, "correct_dc": bool(args.dc and args.kind != "cfloat")},
            "roofline": roof,
            "c5_check": check,
        })
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def quick_leg(torch, dev, local_rank, n, kind_name, nb, threshold, steps, make_input, fc, seq, overlap=False, depth=2, settle_s=0.4):
    """One BASELINE configuration beside the headline, short and settled: `steps` launches of `nb` buffers through the prepared C-ABI
    calls (scn_submit_device / scn_collect for counts and trigger flags), inputs and spectra rotated past the Infinity Cache, HIP
    events on the stream(s) the kernels are launched on.  Returns the leg's object for the line's `configs`."""
    import ctypes as C

    from scanner_amd import Plan, capi

    kind = {"cfloat": capi.KIND_FLOAT_COMPLEX, "int16": capi.KIND_SHORT_COMPLEX, "int8": capi.KIND_BYTE_COMPLEX}[kind_name]
    bps = capi.BYTES_PER_SAMPLE[kind] + 4
    R = max(2, -(-(3 << 29) // (nb * n * bps)))
    raws = [make_input(r) for r in range(R)]
    outs = [torch.empty((nb, n), dtype=torch.float32, device=dev) for _ in range(R)]
    torch.cuda.synchronize()
    flags = capi.OUT_SPECTRUM | capi.OUT_HITS | (capi.PLAN_OVERLAP_SLOTS if overlap else 0)
    plan = Plan(n, FS, threshold, kind=kind, enob=8 if kind_name == "int8" else 12, max_batch=nb, max_hits=nb * max(64, n // 64),
                device_id=local_rank, flags=flags)
    streams = [torch.cuda.ExternalStream(plan.slot_stream_handle(s) if overlap else plan.stream_handle, device=dev) for s in range(depth if overlap else 1)]
    vp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    plan.set_table(fc)  # the centres of a launch are a run of the plan's frequency table: no per-buffer header crosses the boundary
    prep = [(C.c_void_p(raws[r].data_ptr()), nb, 0, vp(seq), C.c_void_p(outs[r].data_ptr())) for r in range(R)]
    pending = [False] * depth
    state = {"launch": 0}

    def run(k_steps):
        for _ in range(k_steps):
            s = state["launch"] % depth
            if pending[s]:
                plan.collect_counts(s)
            plan.submit_prepared_indexed(s, *prep[state["launch"] % R])
            pending[s] = True
            state["launch"] += 1

    def drain():
        for j in range(depth):
            s = (state["launch"] + j) % depth
            if pending[s]:
                plan.collect_counts(s)
                pending[s] = False

    # settle like the headline leg: creating the plan and its inputs lets the GPU fall back towards its idle power state, so run
    # this shape's steps untimed for at least settle_s, and on in chunks until a chunk runs within 2 % of the fastest one seen
    t_s, best = time.perf_counter(), None
    while True:
        tc = time.perf_counter()
        run(200)
        drain()
        torch.cuda.synchronize()
        now = time.perf_counter()
        best = now - tc if best is None else min(best, now - tc)
        if now - t_s >= 2.0 or (now - t_s >= settle_s and now - tc <= 1.02 * best):
            break
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = [torch.cuda.Event(enable_timing=True) for _ in streams]
    e0.record(streams[0])
    for e, st in zip(e1, streams):
        e.record(st)
    run(20)
    drain()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record(streams[0])
    run(steps)
    for e, st in zip(e1, streams):
        e.record(st)
    drain()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ev_ms = max(e0.elapsed_time(e) for e in e1) / steps
    plan.close()
    algo = nb * n * bps
    leg = {"value": round(nb * n * steps / wall / 1e6, 1), "unit": "Msamples/s", "steps": steps, "ms_per_step": round(wall / steps * 1e3, 5),
           "kernel": kernel_name(n, kind_name), "kernel_avg_ms": round(ev_ms, 5),
           "algorithmic_bytes_per_launch": algo, "frac": round(algo / (ev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
           "frac_is": "frac_event", "frac_wall": round(algo * steps / wall / 1e9 / HBM_PEAK_GBS, 4),
           "threshold_db": threshold, "submits_in_flight": depth, "rotating_batches": R}
    leg.update(roofline_from_profile(tracked_profile(n, kind_name, nb), algo))
    return leg


def config_legs(torch, dev, local_rank, steps=200):
    """BASELINE configs C3, the C4 per-GPU share and C5 on the same box in the same run as the C2 headline (N = 1 only): short settled
    legs, reported under `configs`, never in `value`."""
    from scanner_amd import capi, synth

    legs = {}
    try:
        # C3: 8192-pt FFT on int16 interleaved IQ, batch 4096 (on-GPU convert + window + FFT + log-power + threshold)
        n, nb = 8192, 4096
        thr = round(10.0 + 5.0 * np.log10(n / 4096.0), 2)
        _, fc = capi.frequency_table(FS, 0.0, nb * USE_BW * FS, USE_BW, 0.0)
        seq = np.arange(nb, dtype=np.uint64)

        def c3_in(r):
            x = synth.cfloat_batch_torch(n, nb, seed=3 + 1000 * r, device=dev)
            return torch.clamp(torch.round(x * 2047.0), -2048, 2047).to(torch.int16).contiguous()

        legs["c3"] = dict(quick_leg(torch, dev, local_rank, n, "int16", nb, thr, steps, c3_in, fc, seq),
                          workload="C3: 8192-pt FFT on int16 interleaved IQ, batch 4096 resident in HBM, spectrum + hits, one stream, two in flight")
        # C4 per-GPU share: the first 2048 of the 16384 centres (what rank 0 of 8 sweeps), one 4096-pt cfloat buffer per centre, a
        # launch per sweep, every slot on its own stream, three in flight (bench.py --config c4 --centres 2048 --sweeps-per-launch 1)
        n, nb, n_centres = 4096, 2048, 16384
        _, fc = capi.frequency_table(FS, 0.0, n_centres * USE_BW * FS, USE_BW, 0.0, shard=0, n_shards=8)
        seq = np.arange(nb, dtype=np.uint64)
        centres, i0 = synth.c4_emitters(n_centres, n)
        legs["c4_share"] = dict(quick_leg(torch, dev, local_rank, n, "cfloat", nb, 10.0, 3 * steps,
                                          lambda r: synth.c4_shard_torch(n, 0, nb, centres, i0, seed=4 + 1000 * r, device=dev), fc, seq,
                                          overlap=True, depth=3),
                                workload="C4 per-GPU share at 8 GPUs: centres [0, 2048) of the 16384-centre table x 4096-pt cfloat, a launch per "
                                         "sweep, each slot on its own stream (SCN_PLAN_OVERLAP_SLOTS), three in flight: launches overlap, so the "
                                         "step is shorter than one kernel's own begin-to-end time (frac_kernel_rocprof)")
        # ... and the same share in steady state with the ONE collective inside the region: every sweep's hit list gathered through the
        # one-rank communicator (no peers: pack + compaction + the root's wait -- what a rank adds to its own sweep)
        legs["c4_share_gather"] = c4_gather_leg(torch, dev, local_rank, 0, 1, 2048, 3 * steps)
        # C5: 65536-pt 50%-overlap Welch PSD, 32 PSDs per submit, stream resident in HBM
        c5_check = {}
        el, new = welch_leg(torch, dev, local_rank, 5, 32, max(50, steps // 2), 10, check=c5_check)
        roof, ms = welch_roofline(el, max(50, steps // 2), new, 32)
        legs["c5"] = dict({"value": round(new / (ms * 1e-3) / 1e6, 1), "unit": "Msamples/s (new samples)", "steps": max(50, steps // 2),
                           "ms_per_step": round(ms, 5)}, **roof, c5_check=c5_check,
                          workload="C5: 65536-pt 50%-overlap Welch PSD, K=16, 32 PSDs per submit, stream resident in HBM")
    except ParityError:  # a leg whose output is wrong is not a side matter: the run fails
        raise
    except Exception as e:  # a side leg must not cost the run its line
        legs["error"] = f"{type(e).__name__}: {e}"[:300]
    return legs


_RESULT_FD = None


def claim_stdout():
    """The driver reads ONE JSON line from stdout.  Libraries write there too (RCCL prints a version banner to
    stdout when the process group is created), so file descriptor 1 is pointed at stderr for the whole run and
    the result line alone goes to the original stdout (emit)."""
    global _RESULT_FD
    if _RESULT_FD is None:
        sys.stdout.flush()
        _RESULT_FD = os.dup(1)
        os.dup2(2, 1)


GATHER_DEADLINE_S = 120
C4_LEG_DEADLINE_S = 180  # the steady-state leg of configs.c4 at N > 1 (a few seconds of work: generous)


def emit(obj):
    line = (json.dumps(obj) + "\n").encode()
    if _RESULT_FD is None:
        sys.stdout.write(line.decode())
        sys.stdout.flush()
    else:
        os.write(_RESULT_FD, line)


def free_port():
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_self(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: this process becomes a parent that never touches
    the GPU (it does not even import torch), starts one rank per GPU through torch.distributed.run, forwards the single
    JSON line rank 0 prints and exits non-zero if any rank failed."""
    import subprocess

    port = args.master_port or free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this pool (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "1")
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    if proc.returncode != 0 or len(lines) != 1:
        sys.stderr.write(f"bench.py launcher: torch.distributed.run exited {proc.returncode} with {len(lines)} result line(s)\n")
        sys.stderr.write(proc.stdout[-4000:])
        sys.exit(proc.returncode or 1)
    sys.stdout.write(lines[0] + "\n")
    sys.stdout.flush()
    sys.exit(0)


def compare_hit_lists(got, want, power_tol=0.15):
    ok = len(got) == len(want) and all(np.array_equal(got[f], want[f]) for f in ("seq_id", "i", "freq_hz"))
    worst = float(np.abs(got["power_db"].astype(np.float64) - want["power_db"]).max()) if ok and len(got) else None
    return bool(ok and (worst is None or worst <= power_tol)), worst


def dist_setup(args, backend):
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        args.gpus = world
    return dist, rank, local_rank, world


def dry_run_main(args):
    """CPU / gloo: everything of the N > 1 path except the kernels.  Every rank takes its shard of the C4 table, writes
    down the hits the planted emitters of its shard must give (the closed form the GPU run is checked against), the
    lists are gathered (the layout comes from the C-ABI's scn_gather_layout) and rank 0 checks the concatenation."""
    import torch
    import torch.distributed as dist

    from scanner_amd import capi, sweep

    _, rank, _, world = dist_setup(args, "gloo")
    if "RANK" in os.environ:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    c4 = args.config == "c4"
    # c4: the table of --centres entries sharded over the ranks (strong scaling); c2: --batch centres PER rank (weak scaling)
    n, n_centres = 4096, (args.centres if c4 else world * args.batch)
    _, fc_all = capi.frequency_table(FS, 0.0, n_centres * USE_BW * FS, USE_BW, 0.0)
    first, fc = capi.frequency_table(FS, 0.0, n_centres * USE_BW * FS, USE_BW, 0.0, shard=rank, n_shards=world)
    lo, hi = sweep.shard_range(n_centres, rank, world)
    assert (first, first + len(fc)) == (lo, hi) and np.array_equal(fc, fc_all[lo:hi])
    from scanner_amd import synth

    centres, i0 = synth.c4_emitters(n_centres, n)
    want = synth.c4_expected_hits(synth.blackman_harris(n), fc_all, centres, i0, n, FS, args.threshold)
    mine = want[(want["seq_id"] >= lo) & (want["seq_id"] < hi)]
    if 0 <= args.dry_run_quiet_rank < world:  # that shard saw nothing: its records leave the expectation, its (empty) list stays in the gather
        qlo, qhi = sweep.shard_range(n_centres, args.dry_run_quiet_rank, world)
        want = want[(want["seq_id"] < qlo) | (want["seq_id"] >= qhi)]
        if rank == args.dry_run_quiet_rank:
            mine = mine[:0]
    t0 = time.perf_counter()
    with sweep.HitGather(torch.device("cpu")) as g:
        got, per_rank = g.gather(mine)
    elapsed = time.perf_counter() - t0
    # the steady-state form's control flow (c4_gather_leg): a list per sweep, posted two sweeps behind, waited for two sweeps later,
    # over the same message format (HitGather.post / wait on gloo)
    sweeps, steady_ok, steady_lists = 7, True, 0
    with sweep.HitGather(torch.device("cpu")) as g:
        cap = max(1, 2 * len(want))          # (every rank must pass the same cap)
        tickets = []
        for sw in range(sweeps):
            if len(tickets) == 2:
                lst, _ = g.wait(tickets.pop(0))
                if rank == 0:
                    steady_ok &= compare_hit_lists(lst, want, power_tol=0.0)[0]
                    steady_lists += 1
            tickets.append(g.post(mine, cap_per_rank=cap))
        for t in tickets:
            lst, _ = g.wait(t)
            if rank == 0:
                steady_ok &= compare_hit_lists(lst, want, power_tol=0.0)[0]
                steady_lists += 1
    if dist.is_initialized():
        dist.barrier()
    if rank == 0:
        ok, _ = compare_hit_lists(got, want, power_tol=0.0)
        emit({"metric": "Msamples/s (complex samples through convert->window->FFT->dB->threshold)", "value": 0.0,
              "unit": "Msamples/s", "n_gpus": world, "steps": 0, "warmup": 0, "ms_per_step": round(elapsed * 1e3, 3),
              "higher_is_better": True, "scaling": "strong" if c4 else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
              "dry_run": True,
              "config": {"workload": f"DRY RUN on CPU (gloo), no kernels: {'C4' if c4 else 'C2-shaped'} table of {n_centres} centres sharded over {world} "
                                     f"rank(s), planted-emitter hit lists gathered to rank 0", "n": n, "centres": n_centres,
                         "batch_per_gpu": n_centres // world},
              "c4_check": {"expected_hits": int(len(want)), "gathered_hits": int(len(got)), "match": ok,
                           "per_rank": [int(c) for c in per_rank]},
              "gather_every_sweep": {"sweeps": sweeps, "lists_gathered": steady_lists, "match": bool(steady_ok),
                                     "transport": "HitGather.post / wait over gloo: the message format and header reading of scn_gather_post / scn_gather_wait"}})
        if not ok:
            sys.exit(3)
    if dist.is_initialized():
        dist.destroy_process_group()


def abi_bench_legs(n, nb, kind, threshold, steps, depth):
    """The records legs from a C++ process (scanner_amd/host/abi_bench: the C-ABI's step loop with the system HIP runtime).
    Why: a torch process runs on the HIP runtime torch bundles (ROCm 7.0 here), which executes device-to-host copies as blit
    KERNELS; the system runtime (ROCm 7.2) a C++ consumer links -- the reference's ProcessSamples is C++ -- uses an SDMA
    engine, and only that leaves the FFT launch beside the copy alone (scripts/ubench/pcie_beside.hip, d2h_engine.hip).
    Returns {mode/depth: result dict} or {"error": ...}; a child process, started after this one's own legs are done."""
    import subprocess

    exe = os.path.join(ROOT, "scanner_amd", "host", "abi_bench")
    if not os.path.exists(exe):
        return {"error": "scanner_amd/host/abi_bench has not been built (python -m scanner_amd.build)"}
    out = {}
    for mode, d, ho in (("view", depth, 0), ("copy", depth, 0), ("view", 2, 0), ("counts", 2, 0), ("view", depth, 1), ("view", 4, 1), ("landed", depth, 0), ("landed", 4, 1)):
        cmd = [exe, "--n", str(n), "--batch", str(nb), "--kind", kind, "--threshold", str(threshold), "--steps", str(steps),
               "--depth", str(d), "--mode", mode, "--hits-only", str(ho)]
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode != 0 or not line:
                return {"error": f"abi_bench --mode {mode} exited {r.returncode}: {r.stderr[-300:]}"}
            out[f"{mode}_depth{d}" + ("_hits_only" if ho else "")] = json.loads(line[-1])
        except Exception as e:  # a missing leg must not cost the run its line
            return {"error": str(e)[:300]}
    return out


KIND_CPP = {"cfloat": "SCN_K_FLOAT_COMPLEX", "int16": "SCN_K_SHORT_COMPLEX", "int16p": "SCN_K_SHORT", "int8": "SCN_K_BYTE_COMPLEX"}


def kernel_name(n, kind, hits=True, spectrum=True, dc=False, time_domain=False):
    """the kernel's name as rocprofv3 prints it: template arguments <.., KIND, DC, HITS, SPEC> (scn_kernels.hip)"""
    k = KIND_CPP[kind]
    h, sp, d = ("true" if hits else "false"), ("true" if spectrum else "false"), ("true" if dc and kind != "cfloat" else "false")
    if time_domain:
        return f"scn_time_domain_wave_kernel<{k}, {d}>"
    if n in (16, 32, 64, 128):  # 1, 2, 4 or 8 threads per buffer
        return f"scn_fft_tiny_kernel<{n // 16}, {k}, {d}, {h}, {sp}>"
    if n in (256, 512):  # several buffers per workgroup
        return f"scn_fft_small_kernel<{n // 256}, {k}, {d}, {h}, {sp}>"
    if n == 8192:
        return f"scn_fft8k_kernel<{k}, {d}, {h}, {sp}>"
    if n == 16384:  # 32 x 16 x 32, pass 3 in double
        return f"scn_fft16k2_kernel<{k}, {d}, {h}, {sp}>"
    if n in (1024, 2048, 4096):
        return f"scn_fft_kernel<{n // 256}, {k}, {d}, {h}, {sp}>"
    import re

    plans = open(os.path.join(ROOT, "scanner_amd", "csrc", "scn_mixed_plans.h")).read()
    m = re.search(r"X\(%d, (\d+), (\d+), (\d+), (\d+), (\d+), \d+\)" % n, plans)
    if m:  # a mixed-radix fused kernel (scn_mixed.hip); DC removal is a runtime branch there
        return f"scn_fft_mixed_kernel<GeoMixed<{n}, {m.group(1)}, {m.group(2)}, {m.group(3)}, {m.group(4)}, {m.group(5)}>, {k}, {h}, {sp}>"
    m = re.search(r"X\(%d, (\d+), (\d+), (\d+), (\d+), \d+\)" % n, plans)
    if m:  # ... beyond 10000 points: two virtual threads per thread, one in-place exchange
        return f"scn_fft_mixed_big_kernel<GeoMixedBig<{n}, {m.group(1)}, {m.group(2)}, {m.group(3)}, {m.group(4)}>, {k}, {h}, {sp}>"
    if n == 65536:
        return f"scn_big_cols_kernel<{k}, 65536, false> + scn_big_rows_kernel<{h}, {sp}> (four-step 256 x 256, scn_big.hip: the work buffer's round trip is not algorithmic traffic)"
    if n == 32768:
        return f"scn_big_cols_kernel<{k}, 32768, false> + scn_big_rows32k_kernel<{h}, {sp}> (four-step 256 x 128, scn_big.hip: the work buffer's round trip is not algorithmic traffic)"
    return "scn_gen_load_kernel + scn_gen_stage_kernel x log4(n) + scn_gen_finish_kernel (the staged path, scn_generic.hip)"


def shape_key(n, kind, nb, mode="both", dc=False, time_domain=False):
    """key of a launch shape in profiles/measured_shapes.json: size / wire format / buffers per launch, then whatever differs
    from the default plan (spectrum + hits, no DC removal, frequency domain)"""
    return f"{n}/{kind}/{nb}" + ("/td" if time_domain else "" if mode == "both" else "/" + mode) + ("/dc" if dc else "")


def tracked_profile(n, kind, nb, mode="both", dc=False, time_domain=False):
    """Numbers that come from their own rocprofv3 passes (scripts/prof.sh -> profiles/measured_shapes.json): the PMC
    traffic per launch and the kernel-trace average duration; only valid for the launch shape they were collected on."""
    return _tracked(shape_key(n, kind, nb, mode, dc, time_domain))


CLOCK_GHZ, N_SIMD = 2.4, 1024  # MI355X_MICROARCH.md: 256 CUs x 4 SIMDs, 2.4 GHz peak engine clock


def roofline_from_profile(prof, algo_bytes):
    """the fields of a roofline object that come from a tracked profile of THIS build (null otherwise): PMC traffic, the kernel's
    own begin-to-end time under rocprofv3, and -- the second bound, for launches that are not HBM-bound -- valu_frac = the share of
    the chip's VALU issue slots the launch used: SQ_INSTS_VALU (wave instructions) x 4 cycles / (kernel time x clock x SIMDs)"""
    us = prof.get("kernel_avg_us")
    return {"traffic": prof.get("hbm_bytes_per_launch", prof.get("hbm_bytes_per_step")),
            "kernel_avg_us_rocprof": us,
            "frac_kernel_rocprof": round(algo_bytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4) if us else None,
            "valu_frac": prof.get("valu_frac"), "valu_frac_is": "SQ_INSTS_VALU x 4 cycles / (kernel time x 2.4 GHz x 1024 SIMDs)" if prof.get("valu_frac") else None,
            "traffic_source": prof.get("source"), "traffic_build": prof.get("build"), "traffic_stale": prof.get("stale")}


def _tracked(key):
    """An entry of profiles/measured_shapes.json, or -- if it was taken on ANOTHER build of the kernels than the one running
    (scanner_amd.build.source_hash over the HIP sources) -- only the note that it is stale: the live line then carries
    null traffic / rocprof fields instead of another build's numbers."""
    try:
        from scanner_amd import build as _build

        e = json.load(open(os.path.join(ROOT, "profiles", "measured_shapes.json"))).get(key, {})
        if not e:
            return {}
        now = _build.source_hash()
        if e.get("build") != now:
            return {"stale": f"{e.get('source')} was taken on build {e.get('build')}, this is {now}"}
        return e
    except Exception:
        return {}


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:
        return launch_self(args)
    claim_stdout()
    if args.dry_run:
        return dry_run_main(args)
    if args.welch:
        return welch_main(args)
    import torch
    import torch.distributed as dist

    _, rank, local_rank, world = dist_setup(args, "nccl")
    force_dist = world == 1 and "RANK" in os.environ and os.environ.get("SCN_BENCH_FORCE_DIST")  # 1-rank self-test
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the HIP path has no CPU fallback (--dry-run exercises the N>1 plumbing on CPU)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or bool(force_dist)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    from scanner_amd import Plan, capi, synth, sweep

    c4 = args.config == "c4"
    if c4:
        args.n, args.kind = 4096, "cfloat"
    n = args.n
    kind = {"cfloat": capi.KIND_FLOAT_COMPLEX, "int16": capi.KIND_SHORT_COMPLEX, "int16p": capi.KIND_SHORT, "int8": capi.KIND_BYTE_COMPLEX}[args.kind]
    enob = {"cfloat": 12, "int16": 12, "int16p": 12, "int8": 8}[args.kind]
    td = bool(args.time_domain)
    if td:
        args.plan_mode = "hits"  # (nothing but two floats per buffer comes back)
    want_spec, want_hit = args.plan_mode != "hits", args.plan_mode != "spectrum"
    dc = bool(args.dc) and args.kind != "cfloat"
    in_bytes = capi.BYTES_PER_SAMPLE[kind]
    algo_bytes_per_sample = in_bytes + (4 if want_spec else 0)  # raw sample in + one float dB out; hits-only: the samples alone

    # this rank's shard of the frequency table (frequencyTable.cpp:9-37), contiguous range
    if c4:
        n_centres = args.centres
        _, fc_all = capi.frequency_table(FS, 0.0, n_centres * USE_BW * FS, USE_BW, 0.0)
        first, fc = capi.frequency_table(FS, 0.0, n_centres * USE_BW * FS, USE_BW, 0.0, shard=rank, n_shards=world)
        shard = len(fc)                      # buffers this rank sweeps per step
        # A launch is a batch of whatever is queued, not a sweep (the ProcessSamples worker drains up to max_batch messages
        # whatever sweep they belong to): a shard smaller than the C2 batch is launched S sweeps at a time, so the per-launch
        # fixed costs that a 2048-buffer launch cannot hide are paid once per 8192 buffers at every N.
        sweeps_per_launch = max(1, 8192 // shard) if args.sweeps_per_launch == 0 else args.sweeps_per_launch
        nb = min(shard, 8192) if sweeps_per_launch == 1 else shard * sweeps_per_launch  # buffers per launch
    else:
        nb = args.batch
        first, fc = capi.frequency_table(FS, 0.0, world * nb * USE_BW * FS, USE_BW, 0.0, shard=rank, n_shards=world)
        assert len(fc) == nb and first == rank * nb
        shard = nb
        sweeps_per_launch = 1
    S = sweeps_per_launch
    seq = np.arange(first, first + shard, dtype=np.uint64)
    chunks = [(lo, min(lo + nb, shard)) for lo in range(0, shard, nb)]  # launches of one step (S == 1)
    if S > 1:  # headers of a launch of S consecutive sweeps: the same centres, sequence ids running on (messageQueue.h:86)
        fc_launch = np.tile(fc, S)
        seq_launch = np.concatenate([seq + np.uint64(j * (n_centres if c4 else shard)) for j in range(S)])

    # synthetic IQ generated in HBM (seeded per rank and per rotation slot); quantised on device for
    # the int kinds.  R batches in, R spectra out: footprint >= 1.5 GiB >> 256 MiB Infinity Cache.
    step_bytes = shard * n * algo_bytes_per_sample
    R = args.rotate or max(2, -(-(3 << 29) // step_bytes))
    R = -(-R // S) * S  # whole launches
    raws, outs = [], []
    if c4:
        centres, i0 = synth.c4_emitters(n_centres, n)
    for r in range(R):
        if c4:
            x = synth.c4_shard_torch(n, first, shard, centres, i0, seed=4 + 1000 * r, device=dev)
        else:
            x = synth.cfloat_batch_torch(n, nb, seed=2 + rank + 1000 * r, device=dev)
        if kind == capi.KIND_SHORT_COMPLEX:
            raws.append(torch.clamp(torch.round(x * 2047.0), -2048, 2047).to(torch.int16).contiguous())
        elif kind == capi.KIND_SHORT:  # planar: I[n] then Q[n] per buffer
            raws.append(torch.clamp(torch.round(x * 2047.0), -2048, 2047).to(torch.int16).permute(0, 2, 1).contiguous())
        elif kind == capi.KIND_BYTE_COMPLEX:
            raws.append(torch.clamp(torch.round(x * 127.0), -128, 127).to(torch.int8).contiguous())
        else:
            raws.append(x)
        del x
        outs.append(torch.empty((shard, n), dtype=torch.float32, device=dev))
    graws, gouts = [], []
    if S > 1:  # S consecutive rotation slots back to back in memory = one launch; raws / outs become views of them
        for g in range(R // S):
            graws.append(torch.cat(raws[g * S:(g + 1) * S], dim=0).contiguous())
            gouts.append(torch.empty((S * shard, n), dtype=torch.float32, device=dev))
            for j in range(S):
                raws[g * S + j] = graws[g][j * shard:(j + 1) * shard]
                outs[g * S + j] = gouts[g][j * shard:(j + 1) * shard]
    raw = raws[0]
    torch.cuda.synchronize()

    # records the plan keeps per slot / the caller's buffer: ample for this input.  The threshold is an absolute dB figure and
    # the noise floor of a bin grows with N, so beyond the fused sizes a tenth of the bins cross 10 dB
    # (the 10 dB default sits UNDER the noise mean of a 65536-point buffer: 30 % of the bins are hits, 7.7 M records per launch)
    hit_cap = nb * max(64, n // 64) if n <= 16384 else min(nb * (n // 2), 8 << 20)
    # C4 with a launch per sweep: every slot on a stream of its own (SCN_PLAN_OVERLAP_SLOTS) and three slots in the ring.  A
    # shard's launch (2048 buffers at N = 8: 23.8 us of kernel) ends with a third of its workgroups one buffer short of the
    # others, and on one stream the next launch waits for the last of them plus ~5 us of dispatch: 28.6 us per step from a C++
    # caller; overlapped, the next launch's workgroups take the CUs as they come free: 25.9 us (scanner_amd/host/abi_bench,
    # profiles/r04_experiments.md section 3).  The C2-sized launches keep the single-stream plan (see the `overlap` leg).
    c4_overlap = c4 and S == 1 and nb < 8192
    mode_flags = (capi.OUT_SPECTRUM if want_spec else 0) | (capi.OUT_HITS if want_hit else 0)
    flag_names = "|".join(f for f, on in (("SCN_OUT_SPECTRUM", want_spec), ("SCN_OUT_HITS", want_hit)) if on)
    main_flags = mode_flags | (capi.PLAN_OVERLAP_SLOTS if c4_overlap else 0)
    main_depth = 3 if c4_overlap else 2
    plan = Plan(n, FS, args.threshold, kind=kind, enob=enob, correct_dc=dc, max_batch=nb, max_hits=hit_cap, device_id=local_rank, flags=main_flags,
                mode=capi.MODE_TIME_DOMAIN if td else capi.MODE_FREQUENCY_DOMAIN)
    ext = torch.cuda.ExternalStream(plan.stream_handle, device=dev)
    slot_streams = [torch.cuda.ExternalStream(plan.slot_stream_handle(s), device=dev) for s in range(main_depth)] if c4_overlap else [ext]

    def make_loop(pl, want_records, zero_copy=False, spectrum=True, depth=2):
        """step(k): one pass over this rank's batch = len(chunks) launches, double-buffered over two of the plan's slots (`depth`
        of them in the records legs); the results of a slot are collected right before it is reused (counts + trigger flags;
        the ordered records too if asked)"""
        pending = [False] * depth
        state = {"launch": 0, "hits": 0, "acc": 0, "group": 0}
        # the centres of a launch are a run of this rank's frequency table, resident on the GPU (scn_plan_set_table): a submit
        # names its first entry (--per-buffer-centres: the centres travel with every submit, 8 bytes per buffer, as until round 4)
        table = not args.per_buffer_centres
        if table:
            pl.set_table(fc)
        rec_buf = np.zeros(hit_cap, capi.HIT_DTYPE) if want_records else None  # the caller's record buffer, reused
        # counts-only loops with a launch per chunk go through the prepared calls (two ctypes calls per launch, every
        # argument a C value made here once): Python's own cost per step must stay below a 24 us launch
        import ctypes as C
        fast = S == 1 and not want_records and not td
        if fast:
            vp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
            # (ids that are simply the buffers' indices in the launch are not passed at all: the compaction kernel numbers them itself)
            prep = [[(C.c_void_p(raws[r][lo:hi].data_ptr()), hi - lo, lo if table else vp(fc[lo:hi]), None if (first == 0 and lo == 0) else vp(seq[lo:hi]),
                      C.c_void_p(outs[r][lo:hi].data_ptr()) if spectrum else None) for lo, hi in chunks] for r in range(R)]
            submit_prepared = pl.submit_prepared_indexed if table else pl.submit_prepared
            state["keep"] = [fc, seq]

        def collect(s):
            tc0 = time.perf_counter()
            if td:  # time-domain plans: max / min dB and the above-threshold flag per buffer (process.cpp:203-237)
                pl.collect_time_domain(s)
                pending[s] = False
                return
            if zero_copy:  # counts + trigger flags from scn_collect, the records read in place (scn_hits_view)
                pl.collect(s, want_power=False, want_hits=False)
                h = pl.hits_view(s)
            else:
                _, h, _ = pl.collect(s, want_power=False, want_hits=want_records, hits_out=rec_buf)
            state["collect_s"] = state.get("collect_s", 0.0) + time.perf_counter() - tc0
            state["collects"] = state.get("collects", 0) + 1
            if want_records:
                state["hits"] += len(h)
            pending[s] = False

        def flush():  # S > 1: launch the sweeps accumulated so far (a whole group, or what is left at the end)
            a = state["acc"]
            if not a:
                return
            g = state["group"] % len(graws)
            state["group"] += 1
            s = state["launch"] % depth
            state["launch"] += 1
            if pending[s]:
                collect(s)
            pl.submit_device(s, graws[g][:a * shard], a * shard, None if table else fc_launch[:a * shard], seq_launch[:a * shard], sync_producer=False,
                             d_power_db=gouts[g][:a * shard] if spectrum else None, first_index=0 if table else None)
            pending[s] = True
            state["acc"] = 0

        state["flush"] = flush

        def step(k):
            if S > 1:
                state["acc"] += 1
                if state["acc"] == S:
                    flush()
                return
            if fast:
                for a in prep[k % R]:
                    s = state["launch"] % depth
                    state["launch"] += 1
                    if pending[s]:
                        pl.collect_counts(s)
                    submit_prepared(s, *a)
                    pending[s] = True
                return
            for lo, hi in chunks:
                s = state["launch"] % depth
                state["launch"] += 1
                if pending[s]:
                    collect(s)
                pl.submit_device(s, raws[k % R][lo:hi], hi - lo, None if table else fc[lo:hi], None if (first == 0 and lo == 0) else seq[lo:hi], sync_producer=False,
                                 d_power_db=outs[k % R][lo:hi] if spectrum else None, first_index=lo if table else None)
                pending[s] = True

        def drain():
            flush()
            for j in range(depth):  # oldest slot first
                s = (state["launch"] + j) % depth
                if pending[s]:
                    if fast:
                        pl.collect_counts(s)
                        pending[s] = False
                    else:
                        collect(s)

        return step, drain, state

    step, drain, main_state = make_loop(plan, False, spectrum=want_spec, depth=main_depth)

    # The driver times as few as 20 steps (1.5 ms), so everything that is not a step stays out of the region AND out of the gap in
    # front of it: the events exist (a torch event creates its HIP event at the first record), no stream-context switches, no
    # garbage collection inside -- and nothing but the contract's barrier + synchronize between the last warm-up step and t0: the
    # GPU leaves its power state within milliseconds of idling (a gc.collect() in that gap made the first launches of the region
    # 90-110 us instead of 74).
    import gc

    ev0 = torch.cuda.Event(enable_timing=True)
    ev_end = [torch.cuda.Event(enable_timing=True) for _ in slot_streams]  # one per stream that launches (several with overlapped slots)
    ev0.record(ext)
    for e, st_ in zip(ev_end, slot_streams):
        e.record(st_)
    gc.collect()
    gc.disable()
    # settle: the same steps, untimed and reported, until the GPU is out of its idle power state
    # At least args.settle seconds, then on in chunks of 200 steps until a chunk runs within 2 % of the fastest one seen (the first
    # process on a freshly leased box ran its launches at 88 us instead of 74 after the fixed 0.4 s), at most 3 s in all.
    settle_steps = 0
    if args.settle > 0:
        t_settle = time.perf_counter()
        best = None
        while True:
            tc = time.perf_counter()
            for _ in range(200):
                step(settle_steps)
                settle_steps += 1
            drain()
            torch.cuda.synchronize()
            now = time.perf_counter()
            chunk = now - tc
            best = chunk if best is None else min(best, chunk)
            if now - t_settle >= 3.0 or (now - t_settle >= args.settle and chunk <= 1.02 * best):
                break
    for k in range(args.warmup):
        step(k)
    drain()
    launch0 = main_state["launch"]
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record(ext)
    for k in range(args.steps):
        step(k)
    main_state["flush"]()
    for e, st_ in zip(ev_end, slot_streams):
        e.record(st_)
    drain()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0  # this rank's K steps, from the common start; the job's time is the MAX over ranks (below)
    if world > 1:
        dist.barrier()                  # the closing bracket: nobody goes on before everybody is done
    gc.enable()
    launches = main_state["launch"] - launch0
    # average launch-to-launch duration on the stream(s) the kernels are launched on: from the first launch's start to the
    # last stream's last completion
    kernel_ms = max(ev0.elapsed_time(e) for e in ev_end) / launches

    if world > 1:
        tt = torch.tensor([elapsed, kernel_ms], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed, kernel_ms = tt.tolist()

    # the extra legs (reported beside the contract leg, never in `value`) time their own number of steps: a 20-step region is
    # mostly its own start and end, and these legs exist to show steady-state rates
    leg_steps = max(args.steps, 200)
    if args.plan_mode != "both" or td:  # the side legs compare against the default plan: they ride on the default line only
        args.no_records_leg = args.no_overlap_leg = args.no_hits_only_leg = args.no_configs_leg = True

    def timed_leg(pl, want_records, warm, zero_copy=False, spectrum=True, depth=2):
        st, dr, state = make_loop(pl, want_records, zero_copy, spectrum, depth)
        for k in range(max(warm, 50)):
            st(k)
        dr()
        torch.cuda.synchronize()
        state["hits"] = 0
        state["collect_s"], state["collects"] = 0.0, 0
        # launch-to-launch time on the stream the kernels are launched on (a single-stream plan: its own stream)
        lst = torch.cuda.ExternalStream(pl.stream_handle, device=dev)
        le0, le1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if world > 1:
            dist.barrier()
        l0 = state["launch"]
        t2 = time.perf_counter()
        le0.record(lst)
        for k in range(leg_steps):
            st(k)
        state["flush"]()
        le1.record(lst)
        dr()
        torch.cuda.synchronize()
        state["event_ms_per_launch"] = le0.elapsed_time(le1) / max(1, state["launch"] - l0)
        if world > 1:
            dist.barrier()
        el = time.perf_counter() - t2
        if world > 1:
            tt2 = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(tt2, op=dist.ReduceOp.MAX)
            el = tt2.item()
        return el, state

    # Extra leg, reported separately: the same steps with the ordered hit RECORDS fetched every step (what
    # ProcessSamples needs to print the reference's `freq ... power_db ...` lines), plus the host cost of one such
    # scn_collect alone (everything already on the host side of PCIe: counts loop + one memcpy out of pinned memory).
    records = None
    if not args.no_records_leg:
        el3, st3 = timed_leg(plan, True, min(args.warmup, 20), depth=args.records_depth)
        nh = st3["hits"]
        el4, st4 = timed_leg(plan, True, min(args.warmup, 20), zero_copy=True, depth=args.records_depth)
        el3b, _ = timed_leg(plan, True, min(args.warmup, 20))  # two in flight, as the contract leg and rounds 1-2 ran it
        rec_buf = np.zeros(hit_cap, capi.HIT_DTYPE)
        rec_buf[:] = 0  # touched: np.zeros hands out untouched pages, and their first-touch faults would be timed below
        samples_us = []
        for _ in range(7):
            plan.submit_device(0, raws[0][chunks[0][0]:chunks[0][1]], chunks[0][1] - chunks[0][0], fc[:chunks[0][1]], seq[:chunks[0][1]],
                               sync_producer=False)
            plan.wait(0)
            time.sleep(0.002)  # the compaction behind the kernel has finished too
            tc = time.perf_counter()
            _, h1, _ = plan.collect(0, want_power=False, want_hits=True, hits_out=rec_buf)
            samples_us.append((time.perf_counter() - tc) * 1e6)
        collect_us = float(np.median(samples_us))
        records = {"value": round(world * shard * n * leg_steps / el3 / 1e6, 1), "unit": "Msamples/s", "steps": leg_steps,
                   "ms_per_step": round(el3 / leg_steps * 1e3, 5), "hits_per_step": round(nh / max(leg_steps, 1), 1),
                   "collect_with_records_us": round(collect_us, 1), "collect_with_records_us_min": round(min(samples_us), 1),
                   "collect_hits": int(len(h1)), "submits_in_flight": args.records_depth,
                   "two_in_flight": {"value": round(world * shard * n * leg_steps / el3b / 1e6, 1), "ms_per_step": round(el3b / leg_steps * 1e3, 5)},
                   "collect_call_avg_us_in_loop": round(st3["collect_s"] / max(st3["collects"], 1) * 1e6, 1),
                   "zero_copy_view": {"value": round(world * shard * n * leg_steps / el4 / 1e6, 1), "ms_per_step": round(el4 / leg_steps * 1e3, 5),
                                      "collect_plus_view_avg_us_in_loop": round(st4["collect_s"] / max(st4["collects"], 1) * 1e6, 1),
                                      "note": "the same loop reading the records in place through scn_hits_view (no copy into a caller buffer)"},
                   "note": "same steps, scn_collect returns the ordered, completed scn_hit records (built on the GPU) every step; "
                           "collect_with_records_us = the median of 7 such calls on an idle plan whose list is already complete (event wait + "
                           "top-up DMA if the prefetch was short + one memcpy out of pinned memory into the caller's buffer)"}

    # Extra leg, reported separately: the same steps on a plan whose two slots have streams of their own
    # (SCN_PLAN_OVERLAP_SLOTS), so consecutive launches overlap: the next launch's workgroups fill the CUs the
    # finishing launch frees.  Whole-job throughput rises; each kernel's own begin-to-end time -- what the
    # roofline object and rocprofv3 divide by -- grows while it shares the GPU, so it stays out of them.
    overlap = None
    if not args.no_overlap_leg:
        plan2 = Plan(n, FS, args.threshold, kind=kind, enob=enob, max_batch=nb, max_hits=hit_cap, device_id=local_rank,
                     flags=capi.OUT_SPECTRUM | capi.OUT_HITS | capi.PLAN_OVERLAP_SLOTS)
        el2, _ = timed_leg(plan2, False, max(args.warmup, 200))

        overlap = {"value": round(world * shard * n * leg_steps / el2 / 1e6, 1), "unit": "Msamples/s", "steps": leg_steps,
                   "ms_per_step": round(el2 / leg_steps * 1e3, 5),
                   "aggregate_algorithmic_GBs_per_gpu": round(shard * n * algo_bytes_per_sample * leg_steps / el2 / 1e9, 1),
                   "plan_flags": "SCN_OUT_SPECTRUM|SCN_OUT_HITS|SCN_PLAN_OVERLAP_SLOTS",
                   "note": "same steps, slots on two streams so consecutive launches overlap; not used for value/roofline"}
        plan2.close()

    # Hits-only output mode (reported separately, never mixed with spectrum mode): the same steps on a plan
    # without SCN_OUT_SPECTRUM -- no dB spectrum is written, the algorithmic bytes are the raw samples alone.
    hits_only = None
    if not args.no_hits_only_leg:
        plan3 = Plan(n, FS, args.threshold, kind=kind, enob=enob, max_batch=nb, max_hits=hit_cap, device_id=local_rank,
                     flags=capi.OUT_HITS)
        el5, st5 = timed_leg(plan3, False, min(args.warmup, 20), spectrum=False)
        in_bytes = algo_bytes_per_sample - 4
        ho_ms = st5["event_ms_per_launch"]
        ho_algo = nb * n * in_bytes
        hits_only = {"value": round(world * shard * n * leg_steps / el5 / 1e6, 1), "unit": "Msamples/s", "steps": leg_steps,
                     "ms_per_step": round(el5 / leg_steps * 1e3, 5), "algorithmic_bytes_per_sample": in_bytes,
                     "algorithmic_bytes_per_launch": ho_algo,
                     "frac_of_hbm_peak_wall": round(shard * n * in_bytes * leg_steps / el5 / 1e9 / HBM_PEAK_GBS, 4),
                     # the kernel ProcessSamples::ThreadWorker's plans run (scanner_amd/host/process.cpp): by name, its launch-to-launch
                     # time on the plan's stream (HIP events) and, from this build's own rocprofv3 passes, its begin-to-end time,
                     # PMC traffic and VALU issue share
                     "kernel": kernel_name(n, args.kind, True, False), "kernel_avg_ms": round(ho_ms, 5),
                     "frac": round(ho_algo / (ho_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "frac_is": "frac_event",
                     "plan_flags": "SCN_OUT_HITS", "note": "wall clock over the same steps; not used for value/roofline"}
        if rank == 0:
            hits_only.update(roofline_from_profile(tracked_profile(n, args.kind, nb, "hits"), ho_algo))
        plan3.close()

    # What a plain device-to-device copy reaches on this box in this run: torch's copy kernel over buffers no
    # cache can hold, read + write bytes counted.
    copy_gbs = None
    if rank == 0 and not args.no_copy_ref:
        nbytes = 1 << 30
        src = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        dst = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        for _ in range(3):
            dst.copy_(src)
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0.record()
        for _ in range(10):
            dst.copy_(src)
        c1.record()
        torch.cuda.synchronize()
        copy_gbs = 2.0 * nbytes * 10 / (c0.elapsed_time(c1) * 1e-3) / 1e9
        del src, dst

    # final sweep's hit list: collected with records, gathered to rank 0 by the C-ABI's scn_gather_hits
    # (RCCL: count all-gather + grouped send/recv), not timed above
    tg0 = time.perf_counter()
    parts = []
    for j, (lo, hi) in enumerate(chunks):
        if td or not want_hit:  # nothing to gather: a time-domain plan reports two floats per buffer, a spectrum-only plan no records
            parts.append(np.zeros(0, capi.HIT_DTYPE))
            continue
        plan.submit_device(j & 1, raw[lo:hi], hi - lo, fc[lo:hi], seq[lo:hi], sync_producer=False)
        parts.append(plan.collect(j & 1, want_power=False, want_hits=True)[1])
    hits = np.concatenate(parts) if len(parts) > 1 else parts[0]
    gather_info = {"transport": "none (1 rank)"}
    all_hits, per_rank = hits, np.array([len(hits)])
    stuck = False
    if use_dist:
        # The gather runs on a helper thread with a deadline: this is the one place where ranks exchange data through a
        # library the launcher did not set up, and a rank that never arrives must cost the line one field, not the run.
        import threading

        res = {}

        def gather_job():
            try:
                torch.cuda.set_device(dev)  # (the current device is per thread)
                with sweep.HitGather(dev) as g:
                    if len(chunks) == 1 and want_hit and not td:  # the sweep's list is one collected slot: send it from where the compaction kernel left it
                        res["hits"], res["per_rank"] = g.gather_device(plan, 0)
                        how = "scn_gather_hits_device (the slot's device list, no host staging)"
                    else:
                        res["hits"], res["per_rank"] = g.gather(hits)
                        how = "scn_gather_hits"
                res["info"] = {"transport": how + " (RCCL: ncclAllGather {count, status} + grouped ncclSend/ncclRecv to rank 0, one collective; "
                                                  "rank 0 reads the list with scn_gather_fetch)"}
            except Exception as e:
                res["error"] = str(e)[:300]

        th = threading.Thread(target=gather_job, daemon=True)
        th.start()
        th.join(GATHER_DEADLINE_S)
        stuck = th.is_alive()
        # every rank takes the same branch: the fallback is a collective too
        flag = torch.tensor([0 if (stuck or "error" in res) else 1], dtype=torch.int32, device=dev)
        if not stuck:
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if stuck:
            gather_info = {"transport": f"scn_gather_hits did not return within {GATHER_DEADLINE_S} s on rank {rank}; local hits only"}
        elif int(flag.item()) == 1:
            all_hits, per_rank, gather_info = res["hits"], res["per_rank"], res["info"]
        else:  # say so and use the launcher's process group
            fb = sweep.HitGather(None)
            fb.device = dev
            all_hits, per_rank = fb._gather_torch(np.ascontiguousarray(hits, dtype=capi.HIT_DTYPE), 0)
            gather_info = {"transport": "torch.distributed fallback", "scn_gather_hits_error": res.get("error", "failed on another rank")}
    gather_ms = (time.perf_counter() - tg0) * 1e3

    # BASELINE config 4 in steady state -- strong scaling, the hit list of every sweep gathered inside the timed region: a leg of its
    # own (collective: every rank runs it), on the default N > 1 line as configs.c4 beside the weak-scaling value, and on --config c4
    # --gather-every-sweep at any N
    default_shape = not c4 and (n, args.kind, nb) == (4096, "cfloat", 8192) and args.plan_mode == "both" and not td
    c4_steady = None
    leg_stuck = False
    if not stuck and ((args.gather_every_sweep and c4) or ((world > 1 or force_dist) and default_shape and not args.no_configs_leg)):
        # On a helper thread with a deadline when there are peers: the leg's exchange (scn_gather_post: grouped ncclSend / ncclRecv of fixed
        # size) has only ever run with ONE rank on hardware -- no multi-GPU node was available to the builder --, and a rank that never
        # arrives there must cost the line one field, not the run its line.
        import threading

        leg = {}

        def leg_job():
            try:
                torch.cuda.set_device(dev)  # (the current device is per thread)
                leg["out"] = c4_gather_leg(torch, dev, local_rank, rank, world, args.centres, max(200, min(args.steps, 2000)),
                                           sync=dist.barrier if use_dist else None, threshold=args.threshold, sweeps_per_launch=args.sweeps_per_launch)
            except ParityError as e:
                leg["parity"] = str(e)
            except Exception as e:  # a side leg must not cost the run its line
                leg["error"] = f"{type(e).__name__}: {e}"[:300]

        if world > 1:
            th = threading.Thread(target=leg_job, daemon=True)
            th.start()
            th.join(C4_LEG_DEADLINE_S)
            leg_stuck = th.is_alive()
        else:
            leg_job()
        if "parity" in leg:
            print(f"bench.py: {leg['parity']}", file=sys.stderr)
            sys.stdout.flush()
            os._exit(3)
        if leg_stuck:
            c4_steady = {"error": f"the C4 steady-state leg did not return within {C4_LEG_DEADLINE_S} s on rank {rank}: a rank is waiting in the exchange"} if rank == 0 else None
        elif "error" in leg:
            c4_steady = {"error": leg["error"]} if rank == 0 else None
        else:
            c4_steady = leg.get("out")
    if rank == 0:
        sid = all_hits["seq_id"].astype(np.int64)
        order_ok = bool(np.all((np.diff(sid) > 0) | ((np.diff(sid) == 0) & (np.diff(all_hits["i"].astype(np.int64)) > 0)))) if len(all_hits) > 1 else True
        gather_info.update({"per_rank": [int(c) for c in per_rank], "globally_ordered": order_ok})

    # BASELINE.json config this run corresponds to (shape, format); anything else is labelled as what it is
    config_tag = ("C4" if c4 and args.centres == 16384 else "C4 shape" if c4 else
                  "C2" if (n, args.kind, nb) == (4096, "cfloat", 8192) else
                  "C4 per-GPU share (shape only)" if (n, args.kind, nb) == (4096, "cfloat", 2048) else
                  "C3 shape, batched" if (n, args.kind) == (8192, "int16") else
                  "C1 shape, batched" if (n, args.kind) == (1024, "cfloat") else "other")
    samples_per_step = world * shard * n if not c4 else n_centres * n
    value = samples_per_step * args.steps / elapsed / 1e6
    buffers_per_s = samples_per_step / n * args.steps / elapsed
    algo_bytes_per_launch = nb * n * algo_bytes_per_sample
    achieved = algo_bytes_per_launch / (kernel_ms * 1e-3) / 1e9
    wall_launch_ms = elapsed / launches * 1e3

    if rank == 0:
        prof = tracked_profile(n, args.kind, nb, args.plan_mode, dc, td)
        out = {
            "metric": ("Msamples/s (complex samples through convert->max/min dB->threshold: time-domain mode, process.cpp:203-237)" if td else
                       "Msamples/s (complex samples through convert->window->FFT->dB->threshold)"),
            "value": round(value, 1),
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "settle_steps": settle_steps,
            "ms_per_step": round(elapsed / args.steps * 1e3, 5),
            "higher_is_better": True,
            "scaling": "strong" if c4 else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": (f"{config_tag}: full frequency table of {n_centres} centres x {n}-pt FFT+power+threshold, "
                             f"{shard} cfloat buffers per GPU per sweep in launches of {nb}" + (f" (= {S} consecutive sweeps)" if S > 1 else "") +
                             (", each slot on its own stream (SCN_PLAN_OVERLAP_SLOTS), three in flight" if c4_overlap else "") +
                             f", emitters planted on {len(centres)} centres, "
                             if c4 else
                             f"{config_tag}: {n}-pt " + ("time-domain max/min dB + threshold" if td else "FFT+power+threshold") +
                             f", batch {nb} {args.kind} buffers per GPU resident in HBM, " +
                             ("" if args.plan_mode == "both" else f"plan reports {flag_names} only, ") + ("integer-mean DC removal on, " if dc else "")) +
                            f"Blackman-Harris, fs={FS} Hz, threshold {args.threshold} dB; frequency table range-sharded over "
                            f"{world} GPU(s); {settle_steps} untimed settle steps (>= {args.settle} s, until the launch time is steady) before the {args.warmup} warm-up steps",
                "n": n, "batch_per_gpu": shard, "buffers_per_launch": nb, "sample_kind": args.kind,
                "parallelism": f"table-shard x{world}", "plan_flags": flag_names + ("|SCN_PLAN_OVERLAP_SLOTS" if c4_overlap else ""),
                "plan_mode": args.plan_mode, "correct_dc": dc, "time_domain": td,
                "centre_frequencies": "per buffer with every submit" if args.per_buffer_centres else "a run of the plan's GPU-resident frequency table",
                "rotating_batches": R, "footprint_MiB": round(R * step_bytes / 2**20),
            },
            "swept_GHz_per_s": round(buffers_per_s * USE_BW * FS / 1e9, 1),
            "buffers_per_s": round(buffers_per_s, 1),
            "roofline": {
                "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": prof.get("hbm_bytes_per_launch"),
                "kernel": kernel_name(n, args.kind, want_hit, want_spec, dc, td), "kernel_avg_ms": round(kernel_ms, 5),
                "algorithmic_bytes_per_sample": algo_bytes_per_sample,
                "algorithmic_bytes_per_launch": algo_bytes_per_launch,
                # the same bytes over three clocks, side by side: HIP events launch-to-launch on the plan's stream (what
                # `achieved` uses), host wall time per launch, and the kernel's own begin-to-end time under rocprofv3
                # (a separate run: profiles/measured_shapes.json, null if this shape was not profiled)
                "measured_copy_GBs": None if copy_gbs is None else round(copy_gbs, 1),  # torch tensor copy, 1 GiB, read + write bytes
                "frac_of_measured_copy": None if copy_gbs is None else round(achieved / copy_gbs, 4),
                # `frac` IS frac_event: algorithmic bytes / the average launch-to-launch time of the timed steps measured with HIP
                # events on the plan's stream / 8 TB/s.  frac_kernel_rocprof divides by the kernel's own begin-to-end average
                # from profiles/<traffic_source> instead (same build only, else null), frac_wall by host wall time per step.
                "frac_is": "frac_event",
                "frac_event": round(achieved / HBM_PEAK_GBS, 4),
                "frac_wall": round(algo_bytes_per_launch / (wall_launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                # + from this build's own rocprofv3 passes: frac_kernel_rocprof, valu_frac (the second bound), traffic_source / _build / _stale
                **{k: v for k, v in roofline_from_profile(prof, algo_bytes_per_launch).items() if k != "traffic"},
            },
            "threshold_db": args.threshold,
            "hit_density": round(len(all_hits) / max(1, samples_per_step * 0.7485), 6),  # hits per EVALUATED bin (the mask of process.cpp:46-52 keeps 3066 of 4096)
            "final_sweep_hits": int(len(all_hits)),
            "final_sweep_collect_gather_ms": round(gather_ms, 3),
            "gather": gather_info,
        }
        # the records legs again, from a C++ process on the system HIP runtime: the product's consumer (N = 1 only)
        if records is not None and world == 1 and not c4:  # (abi_bench generates the C2 recipe's input, not the C4 sweep)
            legs = abi_bench_legs(n, nb, args.kind, args.threshold, leg_steps, min(args.records_depth, 3))
            if "error" in legs:
                records = {"error": legs["error"], "python_torch_runtime": records}
            else:
                v = legs[f"view_depth{min(args.records_depth, 3)}"]
                c = legs[f"copy_depth{min(args.records_depth, 3)}"]
                records = {
                    "value": v["value"], "unit": "Msamples/s", "steps": v["steps"], "ms_per_step": v["ms_per_step"],
                    "hits_per_step": v["hits_per_step"], "submits_in_flight": v["submits_in_flight"],
                    "path": "C++ consumer (scanner_amd/host/abi_bench, a child process on the system HIP runtime): scn_submit_device, "
                            "scn_collect for counts + trigger flags, the ordered records read in place through scn_hits_view, EVERY record of every list -- what "
                            "ProcessSamples::ThreadWorker does (scanner_amd/host/process.cpp)",
                    "hip_runtime_version": v["hip_runtime_version"],
                    "two_in_flight": {"value": legs["view_depth2"]["value"], "ms_per_step": legs["view_depth2"]["ms_per_step"]},
                    "copied_out_by_scn_collect": {"value": c["value"], "ms_per_step": c["ms_per_step"], "collect_call_avg_us": c["collect_call_avg_us"],
                                                  "note": "the same loop with scn_collect copying the records into the caller's buffer (one core's memcpy "
                                                          "of ~1.6 MB out of pinned memory per step)"},
                    "counts_only_same_harness": {"value": legs["counts_depth2"]["value"], "ms_per_step": legs["counts_depth2"]["ms_per_step"]},
                    # round 4's figures stopped the clock when the list had LANDED in pinned memory (first and last record touched); from
                    # round 5 on `value` has the consumer read every record, as the reference's consumer formats every one
                    "landed_only": {"value": legs[f"landed_depth{min(args.records_depth, 3)}"]["value"], "ms_per_step": legs[f"landed_depth{min(args.records_depth, 3)}"]["ms_per_step"],
                                    "hits_only_plan_four_in_flight": legs["landed_depth4_hits_only"]["value"],
                                    "note": "the same loops with only the first and the last record of each list read: when the DMA has landed, not when the consumer has read it"},
                    "hits_only_plan": {"value": legs[f"view_depth{min(args.records_depth, 3)}_hits_only"]["value"],
                                       "ms_per_step": legs[f"view_depth{min(args.records_depth, 3)}_hits_only"]["ms_per_step"],
                                       "submits_in_flight": legs[f"view_depth{min(args.records_depth, 3)}_hits_only"]["submits_in_flight"],
                                       "four_in_flight": {"value": legs["view_depth4_hits_only"]["value"], "ms_per_step": legs["view_depth4_hits_only"]["ms_per_step"]},
                                       "note": "the same loop on a plan without SCN_OUT_SPECTRUM (the reference prints records, not spectra: "
                                               "what ProcessSamples::ThreadWorker creates); on its own byte count, never mixed into value"},
                    "python_torch_runtime": records,
                    "note": "value = the records loop of a C++ caller.  python_torch_runtime = the same loops driven from this Python "
                            "process, whose HIP runtime (the one torch bundles) runs every device-to-host copy as a blit kernel beside "
                            "the FFT launch instead of on an SDMA engine",
                }
        rc = 0
        if c4:
            want = synth.c4_expected_hits(plan.window(), fc_all, centres, i0, n, FS, args.threshold)
            ok, worst = compare_hit_lists(all_hits, want)
            out["c4_check"] = {"expected_hits": int(len(want)), "gathered_hits": int(len(all_hits)), "match": ok,
                               "max_power_db_diff": worst,
                               "expectation": "closed form: on-bin tone x window DFT, reference frequency arithmetic"}
            rc = 0 if ok else 3
        out["with_hit_records"] = records
        out["overlap"] = overlap
        out["hits_only"] = hits_only
        if world == 1 and not c4 and not args.no_configs_leg and (n, args.kind, nb) == (4096, "cfloat", 8192):
            try:
                out["configs"] = config_legs(torch, dev, local_rank)  # C3, the C4 per-GPU share and C5 beside the C2 headline
            except ParityError as e:
                print(f"bench.py: {e}", file=sys.stderr)
                plan.close()
                sys.exit(3)
        if c4_steady is not None:
            if c4:
                out["gather_every_sweep"] = c4_steady
            else:
                out["configs"] = dict(out.get("configs") or {}, c4=c4_steady)
        # the reference's CPU path beside the GPU number, in the same run on the same box: on rank 0 at every N (north_star; the
        # other ranks wait at the closing barrier meanwhile -- it is outside every timed region)
        if not args.no_cpu_baseline and not td:
            host = raw[: min(shard, 4096)].cpu().numpy()
            okind = {"cfloat": 4, "int16": 3, "int16p": 2, "int8": 1}[args.kind]
            if args.kind == "cfloat":
                host = host.view(np.complex64).reshape(host.shape[0], n)
            args.batch = shard
            out["cpu_baseline"] = cpu_baseline(args, host, okind, enob, args.cpu_seconds, dc)
        else:
            out["cpu_baseline"] = None
        emit(out)
    else:
        rc = 0
    if stuck:  # a thread is still inside the collective: the line is out, leave without the teardown that would wait for it
        sys.stdout.flush()
        os._exit(rc or 4)
    if leg_stuck:  # the same for the side leg -- whose failure is in the line (configs.c4.error) and does not void `value`
        sys.stdout.flush()
        os._exit(rc)
    plan.close()
    if use_dist:
        watchdog = None
        if world > 1:  # (a peer whose side leg hung has left without the teardown: do not wait for it for ever -- the line is out)
            import threading

            watchdog = threading.Timer(60.0, lambda: (sys.stdout.flush(), os._exit(rc)))
            watchdog.daemon = True
            watchdog.start()
        dist.barrier()
        dist.destroy_process_group()
        if watchdog:
            watchdog.cancel()
    if rc:
        sys.exit(rc)


if __name__ == "__main__":
    main()

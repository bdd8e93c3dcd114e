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
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
         --master-port 29500 bench.py --gpus 8
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
    ap.add_argument("--kind", default="cfloat", choices=["cfloat", "int16", "int8"])
    ap.add_argument("--threshold", type=float, default=10.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=6.0,
                    help="approximate wall budget of the CPU baseline (three legs: 1, 2 and 8 threads, ~20 CPU-seconds)")
    ap.add_argument("--rotate", type=int, default=0, help="distinct input/output batches (0: enough for 1.5 GiB)")
    ap.add_argument("--no-overlap-leg", action="store_true",
                    help="skip the extra leg that times the same steps on a plan with SCN_PLAN_OVERLAP_SLOTS "
                         "(reported separately under \"overlap\"; value / roofline are always the single-stream plan)")
    ap.add_argument("--welch", action="store_true", help="BASELINE config C5: streaming 65536-pt 50%%-overlap Welch PSD")
    ap.add_argument("--welch-psd", type=int, default=8, help="PSDs per submit (K=16 segments each)")
    ap.add_argument("--welch-pinned", action="store_true", help="feed from pinned host memory through the captured hipGraph")
    return ap.parse_args()


def cpu_baseline(args, raw_host, kind_oracle, enob, budget_s):
    """Times the oracle (CPU restatement of process.cpp/fft.cpp/utility.cpp, its own FFT --
    NOT FFTW) on a bounded sample of the same buffers.  Checker code used as the reported
    baseline only; never on the product path."""
    from oracle import oracle as O

    O.build()
    O.set_fft_mode(False)  # the float radix-2 FFT: FFTW computes in float too; the double-internal mode is for parity
    n = args.n
    o = O.Oracle(n, FS, args.threshold, kind=kind_oracle, enob=enob)
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


def welch_main(args):
    """C5: every rank processes its own stream (replicas, no collective).  A step = one submit of
    --welch-psd PSDs = (n_psd*16 + 1) * 32768 complex samples, device-resident (rotated over R
    streams past the Infinity Cache) or, with --welch-pinned, staged from pinned host memory
    through the captured hipGraph (PCIe-bound)."""
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
    from scanner_amd import WelchPlan

    N, K, npsd = 65536, 16, args.welch_psd
    plan = WelchPlan(N, K, max_psd=npsd, device_id=local_rank)
    m = plan.samples(npsd)
    new_samples = npsd * K * (N // 2)
    R = args.rotate or max(2, -(-(3 << 29) // (m * 8)))
    g = torch.Generator(device=dev)
    g.manual_seed(5 + rank)
    if args.welch_pinned:
        for s in range(2):
            hb = plan.host_buffer(s)
            hb[:m] = (torch.randn((m, 2), generator=g, device=dev) * 0.05).cpu().numpy().view(np.complex64).reshape(-1)
    else:
        xs = [torch.randn((m, 2), generator=g, device=dev) * 0.05 for _ in range(R)]
        outs = [torch.empty((npsd, N), dtype=torch.float32, device=dev) for _ in range(R)]
    torch.cuda.synchronize()
    pending = [False, False]

    def step(k):
        s = k & 1
        if pending[s]:
            plan.collect(s, want_psd=False)
        if args.welch_pinned:
            plan.submit(s, npsd)
        else:
            plan.submit_device(s, xs[k % R], npsd, d_psd_db=outs[k % R], sync_producer=False)
        pending[s] = True

    def drain():
        for s in (0, 1):
            if pending[s]:
                plan.collect(s, want_psd=False)
                pending[s] = False

    for k in range(args.warmup):
        step(k)
    drain()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = tt.item()
    if rank == 0:
        algo = new_samples * 8 + npsd * N * 4  # 8 B per NEW sample + 4N/K per segment
        ms = elapsed / args.steps * 1e3
        achieved = algo / (ms * 1e-3) / 1e9
        emit({
            "metric": "Msamples/s (new complex samples through 65536-pt 50%-overlap Welch PSD)",
            "value": round(world * new_samples * args.steps / elapsed / 1e6, 1), "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 5),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"C5: 65536-pt 50%-overlap Welch PSD, K=16, {npsd} PSDs per submit, Blackman-Harris, "
                                   f"{'pinned host staging + hipGraph replay' if args.welch_pinned else 'stream resident in HBM'}; "
                                   f"independent stream per GPU (replicas)",
                       "n": N, "segments_per_psd": K, "psd_per_submit": npsd, "pinned": bool(args.welch_pinned)},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                         "kernel": "scn_welch_cols_kernel + scn_welch_rows_kernel (two-pass four-step FFT: the work "
                                   "buffer round trip and the 50% overlap re-read are NOT algorithmic bytes)",
                         "algorithmic_bytes_per_launch": algo},
        })
    plan.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


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


def emit(obj):
    line = (json.dumps(obj) + "\n").encode()
    if _RESULT_FD is None:
        sys.stdout.write(line.decode())
        sys.stdout.flush()
    else:
        os.write(_RESULT_FD, line)


def main():
    args = parse()
    claim_stdout()
    if args.welch:
        return welch_main(args)
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one process per GPU)")
        args.gpus = world
    force_dist = world == 1 and "RANK" in os.environ and os.environ.get("SCN_BENCH_FORCE_DIST")  # 1-rank self-test
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    from scanner_amd import Plan, capi, synth, sweep

    n, nb = args.n, args.batch
    kind = {"cfloat": capi.KIND_FLOAT_COMPLEX, "int16": capi.KIND_SHORT_COMPLEX, "int8": capi.KIND_BYTE_COMPLEX}[args.kind]
    enob = {"cfloat": 12, "int16": 12, "int8": 8}[args.kind]
    in_bytes = capi.BYTES_PER_SAMPLE[kind]
    algo_bytes_per_sample = in_bytes + 4  # raw sample in + one float dB out

    # this rank's shard of the frequency table (frequencyTable.cpp:9-37), contiguous range
    first, fc = capi.frequency_table(FS, 0.0, world * nb * USE_BW * FS, USE_BW, 0.0, shard=rank, n_shards=world)
    assert len(fc) == nb and first == rank * nb
    seq = np.arange(first, first + nb, dtype=np.uint64)

    # synthetic IQ generated in HBM (seeded per rank and per rotation slot); quantised on device for
    # the int kinds.  R batches in, R spectra out: footprint >= 1.5 GiB >> 256 MiB Infinity Cache.
    step_bytes = nb * n * algo_bytes_per_sample
    R = args.rotate or max(2, -(-(3 << 29) // step_bytes))
    raws, outs = [], []
    for r in range(R):
        x = synth.cfloat_batch_torch(n, nb, seed=2 + rank + 1000 * r, device=dev)
        if kind == capi.KIND_SHORT_COMPLEX:
            raws.append(torch.clamp(torch.round(x * 2047.0), -2048, 2047).to(torch.int16).contiguous())
        elif kind == capi.KIND_BYTE_COMPLEX:
            raws.append(torch.clamp(torch.round(x * 127.0), -128, 127).to(torch.int8).contiguous())
        else:
            raws.append(x)
        del x
        outs.append(torch.empty((nb, n), dtype=torch.float32, device=dev))
    raw = raws[0]
    torch.cuda.synchronize()

    plan = Plan(n, FS, args.threshold, kind=kind, enob=enob, max_batch=nb, max_hits=nb * 64, device_id=local_rank)
    ext = torch.cuda.ExternalStream(plan.stream_handle, device=dev)

    pending = [False, False]

    def step(k):
        s = k & 1
        if pending[s]:  # results of the launch two steps ago: per-buffer hit counts + trigger flags
            plan.collect(s, want_power=False, want_hits=False)
        plan.submit_device(s, raws[k % R], nb, fc, seq, sync_producer=False, d_power_db=outs[k % R])
        pending[s] = True

    def drain(k_total):
        for s in ((k_total & 1), ((k_total + 1) & 1)):  # older slot first
            if pending[s]:
                plan.collect(s, want_power=False, want_hits=False)
                pending[s] = False

    # settle: the same steps, untimed and reported, until the GPU is out of its idle power state
    settle_steps = 0
    if args.settle > 0:
        t_settle = time.perf_counter()
        while time.perf_counter() - t_settle < args.settle or (settle_steps & 1):
            step(settle_steps)
            settle_steps += 1
        drain(settle_steps)
        torch.cuda.synchronize()
    for k in range(args.warmup):
        step(k)
    drain(args.warmup)
    torch.cuda.synchronize()

    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(ext):
        ev0.record(ext)
    for k in range(args.steps):
        step(k)
    with torch.cuda.stream(ext):
        ev1.record(ext)
    drain(args.steps)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / args.steps  # average launch-to-launch duration on the plan's stream

    if world > 1:
        tt = torch.tensor([elapsed, kernel_ms], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed, kernel_ms = tt.tolist()

    # Extra leg, reported separately: the same steps on a plan whose two slots have streams of their own
    # (SCN_PLAN_OVERLAP_SLOTS), so consecutive launches overlap: the next launch's workgroups fill the CUs the
    # finishing launch frees.  Whole-job throughput rises; each kernel's own begin-to-end time -- what the
    # roofline object and rocprofv3 divide by -- grows while it shares the GPU, so it stays out of them.
    overlap = None
    if not args.no_overlap_leg:
        plan2 = Plan(n, FS, args.threshold, kind=kind, enob=enob, max_batch=nb, max_hits=nb * 64, device_id=local_rank,
                     flags=capi.OUT_SPECTRUM | capi.OUT_HITS | capi.PLAN_OVERLAP_SLOTS)
        pend2 = [False, False]

        def step2(k):
            s = k & 1
            if pend2[s]:
                plan2.collect(s, want_power=False, want_hits=False)
            plan2.submit_device(s, raws[k % R], nb, fc, seq, sync_producer=False, d_power_db=outs[k % R])
            pend2[s] = True

        def drain2():
            for s in (0, 1):
                if pend2[s]:
                    plan2.collect(s, want_power=False, want_hits=False)
                    pend2[s] = False

        for k in range(max(args.warmup, 200)):
            step2(k)
        drain2()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t2 = time.perf_counter()
        for k in range(args.steps):
            step2(k)
        drain2()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        el2 = time.perf_counter() - t2
        if world > 1:
            tt = torch.tensor([el2], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el2 = tt.item()
        overlap = {"value": round(world * nb * n * args.steps / el2 / 1e6, 1), "unit": "Msamples/s",
                   "ms_per_step": round(el2 / args.steps * 1e3, 5),
                   "aggregate_algorithmic_GBs_per_gpu": round(nb * n * algo_bytes_per_sample * args.steps / el2 / 1e9, 1),
                   "plan_flags": "SCN_OUT_SPECTRUM|SCN_OUT_HITS|SCN_PLAN_OVERLAP_SLOTS",
                   "note": "same steps, slots on two streams so consecutive launches overlap; not used for value/roofline"}
        plan2.close()

    # final sweep's hit list: collected with records, gathered to rank 0 over RCCL (not timed above)
    plan.submit_device(0, raw, nb, fc, seq, sync_producer=False)
    tg0 = time.perf_counter()
    _, hits, trig = plan.collect(0, want_power=False, want_hits=True, hit_cap=nb * 64)
    all_hits = sweep.gather_hits(hits, dev) if (world > 1 or force_dist) else hits
    gather_ms = (time.perf_counter() - tg0) * 1e3

    # BASELINE.json config this run corresponds to (shape, format); anything else is labelled as what it is
    config_tag = ("C2" if (n, args.kind, nb) == (4096, "cfloat", 8192) else
                  "C4 per-GPU share" if (n, args.kind, nb) == (4096, "cfloat", 2048) else
                  "C3 shape, batched" if (n, args.kind) == (8192, "int16") else
                  "C1 shape, batched" if (n, args.kind) == (1024, "cfloat") else "other")
    samples_per_step = world * nb * n
    value = samples_per_step * args.steps / elapsed / 1e6
    buffers_per_s = world * nb * args.steps / elapsed
    algo_bytes_per_launch = nb * n * algo_bytes_per_sample
    achieved = algo_bytes_per_launch / (kernel_ms * 1e-3) / 1e9

    if rank == 0:
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc):
            # counters come from their own rocprofv3 passes (scripts/prof.sh); only valid for the launch shape they
            # were collected on
            try:
                j = json.load(open(pmc))
                if (j.get("n"), j.get("sample_kind"), j.get("batch_per_gpu")) == (n, args.kind, nb):
                    traffic = j.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "Msamples/s (complex samples through convert->window->FFT->dB->threshold)",
            "value": round(value, 1),
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "settle_steps": settle_steps,
            "ms_per_step": round(elapsed / args.steps * 1e3, 5),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{config_tag}: {n}-pt FFT+power+threshold, batch {nb} {args.kind} buffers per GPU resident in HBM, "
                            f"Blackman-Harris, fs={FS} Hz, threshold {args.threshold} dB; frequency table "
                            f"range-sharded over {world} GPU(s)",
                "n": n, "batch_per_gpu": nb, "sample_kind": args.kind, "parallelism": f"table-shard x{world}",
                "rotating_batches": R, "footprint_MiB": round(R * step_bytes / 2**20),
            },
            "swept_GHz_per_s": round(buffers_per_s * USE_BW * FS / 1e9, 1),
            "buffers_per_s": round(buffers_per_s, 1),
            "roofline": {
                "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "kernel": "scn_fft_kernel<M=n/256, kind, dc, hits>", "kernel_avg_ms": round(kernel_ms, 5),
                "algorithmic_bytes_per_sample": algo_bytes_per_sample,
                "algorithmic_bytes_per_launch": algo_bytes_per_launch,
            },
            "final_sweep_hits": int(len(all_hits)),
            "final_sweep_collect_gather_ms": round(gather_ms, 3),
        }
        out["overlap"] = overlap
        if not args.no_cpu_baseline and world == 1:
            host = raw[: min(nb, 4096)].cpu().numpy()
            okind = {"cfloat": 4, "int16": 3, "int8": 1}[args.kind]
            if args.kind == "cfloat":
                host = host.view(np.complex64).reshape(host.shape[0], n)
            out["cpu_baseline"] = cpu_baseline(args, host, okind, enob, args.cpu_seconds)
        elif world == 1:
            out["cpu_baseline"] = None
        emit(out)
    plan.close()
    if world > 1 or force_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

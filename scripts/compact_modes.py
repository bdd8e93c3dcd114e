"""Where should the ordered hit list be built?  Times the double-buffered step loop of bench.py (one launch per step,
inputs/outputs rotated past the Infinity Cache) per launch shape with the compaction (scn_hits.hip) on the side stream
(SCN_EXP_COMPACT=0), in order on the compute stream (1), or on demand at collect time (2); each with counts-only collects
and with the records fetched every step; optionally with N CUs left to the side stream (SCN_EXP_RESERVE_CUS).
   python scripts/compact_modes.py [steps]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scanner_amd import Plan, capi, synth

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
dev = torch.device("cuda", 0)
FS = 8000000
shapes = [(4096, "cfloat", 8192), (4096, "int16", 8192), (8192, "int16", 4096), (8192, "cfloat", 4096), (1024, "cfloat", 32768), (4096, "cfloat", 2048)]
if len(sys.argv) > 2:
    shapes = [s for s in shapes if f"{s[0]}/{s[1]}/{s[2]}" in sys.argv[2:]]
for n, kind_name, nb in shapes:
    kind = {"cfloat": capi.KIND_FLOAT_COMPLEX, "int16": capi.KIND_SHORT_COMPLEX}[kind_name]
    bps = capi.BYTES_PER_SAMPLE[kind] + 4
    R = max(2, -(-(3 << 29) // (nb * n * bps)))
    raws, outs = [], []
    for r in range(R):
        x = synth.cfloat_batch_torch(n, nb, seed=2 + 1000 * r, device=dev)
        raws.append(torch.clamp(torch.round(x * 2047.0), -2048, 2047).to(torch.int16).contiguous() if kind_name == "int16" else x)
        outs.append(torch.empty((nb, n), dtype=torch.float32, device=dev))
    fc = 3e6 + 6e6 * np.arange(nb)
    torch.cuda.synchronize()
    for mode, reserve in ((0, 0), (1, 0), (2, 0), (3, 0)):
        os.environ["SCN_EXP_COMPACT"] = str(mode)
        os.environ["SCN_EXP_RESERVE_CUS"] = str(reserve)
        res = []
        for records in (False, True):
            plan = Plan(n, FS, 10.0, kind=kind, enob=12, max_batch=nb, max_hits=nb * 64)
            pend = [False, False]
            nh = 0
            buf = np.zeros(nb * 64, capi.HIT_DTYPE) if records else None

            def step(k):
                global nh
                s = k & 1
                if pend[s]:
                    _, h, _ = plan.collect(s, want_power=False, want_hits=records, hits_out=buf)
                    nh = len(h) if records else 0
                plan.submit_device(s, raws[k % R], nb, fc, None, sync_producer=False, d_power_db=outs[k % R])
                pend[s] = True

            def drain():
                for s in (0, 1):
                    if pend[s]:
                        plan.collect(s, want_power=False, want_hits=records, hits_out=buf)
                        pend[s] = False

            for k in range(300):
                step(k)
            drain()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(steps):
                step(k)
            drain()
            torch.cuda.synchronize()
            res.append((time.perf_counter() - t0) / steps * 1e6)
            plan.close()
        print(f"{n:5d} {kind_name:6s} batch {nb:5d}  compact={mode} reserve={reserve:2d}:  counts-only {res[0]:7.2f} us/step   with records {res[1]:7.2f} us/step  ({nh} hits)", flush=True)
    del raws, outs
    torch.cuda.empty_cache()

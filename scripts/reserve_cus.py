"""Does leaving a few CUs' worth of workgroup slots free help the launches whose workgroups fill every VGPR (8192 points:
254 VGPRs x 2 waves per SIMD)?  The 16-32 KiB count copy behind each launch is a blit KERNEL on this ROCm, and it
cannot start beside such a launch; the host then learns the counts late and submits the launch after next late.
   python scripts/reserve_cus.py [steps]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scanner_amd import Plan, capi, synth

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 600
dev = torch.device("cuda", 0)
for n, kind_name, nb in ((8192, "int16", 4096), (8192, "cfloat", 4096), (4096, "cfloat", 8192), (4096, "int16", 8192)):
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
    for reserve in (0, 2, 4, 8, 16, 0):
        os.environ["SCN_EXP_RESERVE_CUS"] = str(reserve)
        plan = Plan(n, 8000000, 10.0, kind=kind, enob=12, max_batch=nb, max_hits=nb * 64)
        pend = [False, False]

        def step(k):
            s = k & 1
            if pend[s]:
                plan.collect(s, want_power=False, want_hits=False)
            plan.submit_device(s, raws[k % R], nb, fc, None, sync_producer=False, d_power_db=outs[k % R])
            pend[s] = True

        for k in range(400):
            step(k)
        for s in (0, 1):
            plan.collect(s, want_power=False, want_hits=False)
        pend = [False, False]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            step(k)
        for s in (0, 1):
            plan.collect(s, want_power=False, want_hits=False)
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / steps * 1e6
        plan.close()
        print(f"{n:5d} {kind_name:6s} batch {nb:5d} reserve {reserve:2d} CUs: {us:7.2f} us/step  {nb*n/us/1e3:7.1f} Gsamples/s", flush=True)
    del raws, outs
    torch.cuda.empty_cache()

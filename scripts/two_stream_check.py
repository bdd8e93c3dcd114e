"""Does overlapping consecutive launches (two streams) recover the per-launch fixed cost?  Two plans = two streams."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from scanner_amd import Plan, capi, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
nb = 8192 * 4096 // n
R = 4
dev = torch.device('cuda', 0)
xs = [synth.cfloat_batch_torch(n, nb, seed=2 + r, device=dev) for r in range(R)]
outs = [torch.empty((nb, n), dtype=torch.float32, device=dev) for r in range(R)]
fc = 3e6 + 6e6 * np.arange(nb)
for flags in (1, 3):
    plans = [Plan(n, 8000000, 10.0, max_batch=nb, max_hits=nb * 64, flags=flags) for _ in range(2)]
    for mode in ("one stream", "two streams"):
        def step(k):
            p = plans[k & 1] if mode == "two streams" else plans[0]
            s = (k >> 1) & 1 if mode == "two streams" else k & 1
            if pend.get((id(p), s)): p.collect(s, False, False)
            p.submit_device(s, xs[k % R], nb, fc, sync_producer=False, d_power_db=outs[k % R]); pend[(id(p), s)] = True
        res = []
        for rnd in range(5):
            pend = {}
            for k in range(400): step(k)
            for p in plans:
                for s in (0, 1):
                    if pend.get((id(p), s)): p.collect(s, False, False)
            pend = {}
            torch.cuda.synchronize()
            K = 400
            t0 = time.perf_counter()
            for k in range(K): step(k)
            for p in plans:
                for s in (0, 1):
                    if pend.get((id(p), s)): p.collect(s, False, False)
            torch.cuda.synchronize()
            res.append((time.perf_counter() - t0) / K * 1e6)
        r = sorted(res)
        print(f"n={n} flags={flags} {mode:12s}: median {r[2]:7.2f} us per launch (wall), min {r[0]:7.2f}   {nb*n/r[2]/1e3:6.1f} Gsamples/s")
    for p in plans: p.close()

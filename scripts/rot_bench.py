"""Pipelined launch timing with R distinct input batches rotated per step (defeats the 256 MiB
Infinity Cache keeping a re-read input resident)."""
import sys, os, numpy as np, torch
sys.path.insert(0, '.')
from scanner_amd import Plan, capi, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
nb = 8192 * 4096 // n
dev = torch.device('cuda', 0)
tag = os.path.basename(os.environ.get("SCN_LIB", "default")).replace("lib_", "").replace(".so", "")
for R in (4,):
    xs = [synth.cfloat_batch_torch(n, nb, seed=2 + r, device=dev) for r in range(R)]
    outs = [torch.empty((nb, n), dtype=torch.float32, device=dev) for r in range(R)]
    fc = 3e6 + 6e6 * np.arange(nb)
    for flags, thr in ((3, 10.0), (3, 1e9), (1, 10.0)):
        p = Plan(n, 8000000, thr, max_batch=nb, max_hits=nb * 64, flags=flags)
        ext = torch.cuda.ExternalStream(p.stream_handle, device=dev)
        res = []
        for rnd in range(5):
            K = 24
            pend = [False, False]
            for k in range(4):
                p.submit_device(k & 1, xs[k % R], nb, fc, sync_producer=False, d_power_db=outs[k % R]); p.collect(k & 1, False, False)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(ext)
            for k in range(K):
                s = k & 1
                if pend[s]: p.collect(s, False, False)
                p.submit_device(s, xs[k % R], nb, fc, sync_producer=False, d_power_db=outs[k % R]); pend[s] = True
            e1.record(ext)
            for s in (0, 1):
                if pend[s]: p.collect(s, False, False)
            torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) / K * 1e3)
        r = sorted(res)
        print(f"{tag:10s} n={n} R={R} (in+out rotated) flags={flags} thr={thr}: median {r[2]:7.2f} us  min {r[0]:7.2f}  {nb*n*12/r[2]/1e6:5.2f} TB/s")
        p.close()
    del xs
    torch.cuda.empty_cache()

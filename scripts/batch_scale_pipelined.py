"""us per launch vs batch size, launches pipelined over both slots (what bench.py does), inputs/outputs rotated.
   python scripts/batch_scale_pipelined.py <n> <kind: cfloat|int16> [flags]"""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from scanner_amd import Plan, capi, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
kindname = sys.argv[2] if len(sys.argv) > 2 else "cfloat"
flags = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device('cuda', 0)
kind = {"cfloat": capi.KIND_FLOAT_COMPLEX, "int16": capi.KIND_SHORT_COMPLEX}[kindname]
big = 16384 * 4096 // n
R = 3
xs = []
for r in range(R):
    x = synth.cfloat_batch_torch(n, big, seed=2 + r, device=dev)
    if kindname == "int16":
        x = torch.clamp(torch.round(x * 2047.0), -2048, 2047).to(torch.int16).contiguous()
    xs.append(x)
outs = [torch.empty((big, n), dtype=torch.float32, device=dev) for r in range(R)]
prev = None
for nb in (big // 8, big // 4, big // 2, big):
    p = Plan(n, 8000000, 10.0, kind=kind, enob=12, max_batch=nb, max_hits=nb * 64, flags=flags)
    ext = torch.cuda.ExternalStream(p.stream_handle, device=dev)
    fc = 3e6 + 6e6 * np.arange(nb)
    res = []
    for rnd in range(5):
        K = 40; pend = [False, False]
        for k in range(6):
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
    t = sorted(res)[2]
    extra = f"   marginal {(t - prev[1]) / (nb - prev[0]) * 1e3:7.3f} ns/buffer" if prev else ""
    print(f"n={n} {kindname} flags={flags} nb={nb:6d}  {t:8.2f} us/launch  {t/nb*1e3:7.3f} ns/buffer  {nb*n/t/1e3:6.1f} Gsamples/s{extra}")
    prev = (nb, t)
    p.close()

"""Launch duration per rotating input batch (events after every submit): does the time depend on the data (hit count)?
   python scripts/per_batch_times.py <n> <cfloat|int16> [threshold]"""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from scanner_amd import Plan, capi, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
kindname = sys.argv[2] if len(sys.argv) > 2 else "int16"
thr = float(sys.argv[3]) if len(sys.argv) > 3 else 10.0
nb = 8192 * 4096 // n
R = 6
dev = torch.device('cuda', 0)
kind = {"cfloat": capi.KIND_FLOAT_COMPLEX, "int16": capi.KIND_SHORT_COMPLEX}[kindname]
xs = []
for r in range(R):
    x = synth.cfloat_batch_torch(n, nb, seed=2 + 1000 * r, device=dev)
    if kindname == "int16":
        x = torch.clamp(torch.round(x * 2047.0), -2048, 2047).to(torch.int16).contiguous()
    xs.append(x)
outs = [torch.empty((nb, n), dtype=torch.float32, device=dev) for r in range(R)]
fc = 3e6 + 6e6 * np.arange(nb)
p = Plan(n, 8000000, thr, kind=kind, enob=12, max_batch=nb, max_hits=nb * 64)
ext = torch.cuda.ExternalStream(p.stream_handle, device=dev)
K = 600
for k in range(K):   # settle
    p.submit_device(k & 1, xs[k % R], nb, fc, sync_producer=False, d_power_db=outs[k % R]); p.collect(k & 1, False, False)
K = 240
evs = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
pend = [False, False]; hits = {}
torch.cuda.synchronize()
evs[0].record(ext)
order = []
for k in range(K):
    s = k & 1
    if pend[s]:
        p.collect(s, False, False); hits[order[-2]] = p.last_n_hits
    p.submit_device(s, xs[k % R], nb, fc, sync_producer=False, d_power_db=outs[k % R]); pend[s] = True
    order.append(k % R)
    evs[k + 1].record(ext)
for s in (0, 1):
    if pend[s]: p.collect(s, False, False)
torch.cuda.synchronize()
d = np.array([evs[k].elapsed_time(evs[k + 1]) * 1e3 for k in range(K)])
for r in range(R):
    print(f"n={n} {kindname} thr={thr}: batch {r}: {d[r::R].mean():6.1f} us per launch (min {d[r::R].min():5.1f})   hits in batch {hits.get(r)}")
print(f"overall {d.mean():.1f} us")

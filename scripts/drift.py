"""us per launch in chunks of 100 pipelined launches over a long run (clock / power-management transients)."""
import sys, numpy as np, torch, time
sys.path.insert(0, '.')
from scanner_amd import Plan, capi, synth
n, nb, R = 4096, 8192, 4
chunks = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device('cuda', 0)
xs = [synth.cfloat_batch_torch(n, nb, seed=2 + 1000 * r, device=dev) for r in range(R)]
outs = [torch.empty((nb, n), dtype=torch.float32, device=dev) for r in range(R)]
fc = 3e6 + 6e6 * np.arange(nb)
p = Plan(n, 8000000, 10.0, max_batch=nb, max_hits=nb * 64, flags=3)
ext = torch.cuda.ExternalStream(p.stream_handle, device=dev)
torch.cuda.synchronize(); time.sleep(1.0)   # start from an idle GPU
evs = [torch.cuda.Event(enable_timing=True) for _ in range(chunks + 1)]
pend = [False, False]
k = 0
evs[0].record(ext)
for c in range(chunks):
    for _ in range(100):
        s = k & 1
        if pend[s]: p.collect(s, False, False)
        p.submit_device(s, xs[k % R], nb, fc, sync_producer=False, d_power_db=outs[k % R]); pend[s] = True
        k += 1
    evs[c + 1].record(ext)
for s in (0, 1):
    if pend[s]: p.collect(s, False, False)
torch.cuda.synchronize()
d = np.array([evs[c].elapsed_time(evs[c + 1]) * 10 for c in range(chunks)])   # us per launch
print("us/launch per chunk of 100 launches:")
for i in range(0, chunks, 20):
    print(f"  launches {i*100:6d}+: " + " ".join(f"{x:5.1f}" for x in d[i:i + 20]))
print(f"mean {d.mean():.2f}  median {np.median(d):.2f}  first 10 chunks {d[:10].mean():.2f}  last 50 chunks {d[-50:].mean():.2f}")

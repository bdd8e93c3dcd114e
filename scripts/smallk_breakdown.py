"""Where the fixed cost of a SHORT timed region goes (the driver times 20 steps = 1.5 ms): host stamps around the same
double-buffered loop bench.py runs, GPU time from events on the plan's stream."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from scanner_amd import Plan, capi, synth
n, nb, R, K = 4096, 8192, 4, 20
dev = torch.device('cuda', 0)
xs = [synth.cfloat_batch_torch(n, nb, seed=2 + 1000 * r, device=dev) for r in range(R)]
outs = [torch.empty((nb, n), dtype=torch.float32, device=dev) for r in range(R)]
fc = 3e6 + 6e6 * np.arange(nb); seq = np.arange(nb, dtype=np.uint64)
p = Plan(n, 8000000, 10.0, max_batch=nb, max_hits=nb * 64)
ext = torch.cuda.ExternalStream(p.stream_handle, device=dev)
def loop(k0, K, stamps=None):
    pend = [False, False]
    for k in range(K):
        s = k & 1
        if pend[s]: p.collect(s, False, False)
        p.submit_device(s, xs[(k0 + k) % R], nb, fc, seq, sync_producer=False, d_power_db=outs[(k0 + k) % R]); pend[s] = True
        if stamps is not None and k == 0: stamps.append(time.perf_counter())
    if stamps is not None: stamps.append(time.perf_counter())
    for s in ((K & 1), ((K + 1) & 1)):
        if pend[s]:
            p.collect(s, False, False)
            if stamps is not None: stamps.append(time.perf_counter())
t_end = time.time() + 0.4
while time.time() < t_end: loop(0, 50)
torch.cuda.synchronize()
rows = []
for rep in range(12):
    loop(0, 5); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st = []
    torch.cuda.synchronize(); t0 = time.perf_counter()
    e0.record(ext)
    loop(rep, K, st)
    e1.record(ext)   # (after the drain here: both collects have returned, the stream is empty)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    rows.append([(st[0] - t0) * 1e6, (st[1] - t0) * 1e6, (st[2] - st[1]) * 1e6, (st[3] - st[2]) * 1e6, (t1 - st[3]) * 1e6, (t1 - t0) * 1e6])
r = np.array(rows)
print("us: first submit returned | all 20 submitted | collect A | collect B | final sync | total wall   (per-step wall)")
for x in r: print("   " + "  ".join(f"{v:8.1f}" for v in x) + f"   ({x[5] / K:.2f})")
print("median:", "  ".join(f"{v:8.1f}" for v in np.median(r, axis=0)), f"  ({np.median(r[:, 5]) / K:.2f} us per step; steady state ~73.5)")
p.close()

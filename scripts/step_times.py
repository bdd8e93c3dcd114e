"""Per-launch durations (event after every submit) to look for periodic slow launches."""
import sys, os, numpy as np, torch
sys.path.insert(0, '.')
from scanner_amd import Plan, capi, synth
n, nb, R = 4096, 8192, 4
dev = torch.device('cuda', 0)
xs = [synth.cfloat_batch_torch(n, nb, seed=2 + 1000 * r, device=dev) for r in range(R)]
outs = [torch.empty((nb, n), dtype=torch.float32, device=dev) for r in range(R)]
fc = 3e6 + 6e6 * np.arange(nb)
for flags, thr in ((3, 10.0), (3, 1e9), (1, 10.0)):
    p = Plan(n, 8000000, thr, max_batch=nb, max_hits=nb * 64, flags=flags)
    ext = torch.cuda.ExternalStream(p.stream_handle, device=dev)
    K = 48
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
    pend = [False, False]
    for k in range(4):
        p.submit_device(k & 1, xs[k % R], nb, fc, sync_producer=False, d_power_db=outs[k % R]); p.collect(k & 1, False, False)
    torch.cuda.synchronize()
    evs[0].record(ext)
    hits = []
    for k in range(K):
        s = k & 1
        if pend[s]:
            p.collect(s, False, False); hits.append(p.last_n_hits)
        p.submit_device(s, xs[k % R], nb, fc, sync_producer=False, d_power_db=outs[k % R]); pend[s] = True
        evs[k + 1].record(ext)
    for s in (0, 1):
        if pend[s]: p.collect(s, False, False)
    torch.cuda.synchronize()
    d = [evs[k].elapsed_time(evs[k + 1]) * 1e3 for k in range(K)]
    print(f"flags={flags} thr={thr}: mean {np.mean(d):.1f} us; per step:", " ".join(f"{x:.0f}" for x in d))
    print("   hits per step:", hits[:8])
    p.close()

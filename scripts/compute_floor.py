"""Compute floor on REAL data: stores dropped (SCN_EXP_NO_STORES build), input small enough to stay in L2."""
import sys, os, numpy as np, torch
sys.path.insert(0, '.')
from scanner_amd import Plan, capi, synth
n = 4096
dev = torch.device('cuda', 0)
tag = os.path.basename(os.environ.get("SCN_LIB", "default")).replace("lib_", "").replace(".so", "")
for nb, zero in ((3072, False), (3072, True), (6144, False), (6144, True)):
    x = synth.cfloat_batch_torch(n, nb, seed=2, device=dev)
    if zero: x.zero_()
    fc = np.zeros(nb)
    p = Plan(n, 8000000, 10.0, max_batch=nb, flags=1)
    ext = torch.cuda.ExternalStream(p.stream_handle, device=dev)
    res = []
    for rnd in range(5):
        K = 40
        for k in range(3):
            p.submit_device(0, x, nb, fc, sync_producer=False); p.collect(0, False, False)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        pend = [False, False]
        e0.record(ext)
        for k in range(K):
            s = k & 1
            if pend[s]: p.collect(s, False, False)
            p.submit_device(s, x, nb, fc, sync_producer=False); pend[s] = True
        e1.record(ext)
        for s in (0, 1):
            if pend[s]: p.collect(s, False, False)
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / K * 1e3)
    r = sorted(res)[2]
    print(f"{tag:10s} nb={nb} zero={zero}: {r:7.2f} us/launch = {r/nb*8192:7.2f} us per 8192 buffers")
    p.close()

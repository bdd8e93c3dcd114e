"""Quick A/B timing of plan variants on one GPU (interleaved rounds in one process)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from scanner_amd import Plan, capi, synth

n, nb = 4096, 8192
dev = torch.device('cuda', 0)
x = synth.cfloat_batch_torch(n, nb, seed=2, device=dev)
raw16 = torch.clamp(torch.round(x * 2047.0), -2048, 2047).to(torch.int16).contiguous()
raw8 = torch.clamp(torch.round(x * 127.0), -128, 127).to(torch.int8).contiguous()
fc = 3e6 + 6e6 * np.arange(nb)
variants = {
    "cfloat spec+hits": dict(kind=capi.KIND_FLOAT_COMPLEX, flags=3, raw=x, thr=10.0),
    "cfloat spec only": dict(kind=capi.KIND_FLOAT_COMPLEX, flags=1, raw=x, thr=10.0),
    "cfloat hits only": dict(kind=capi.KIND_FLOAT_COMPLEX, flags=2, raw=x, thr=10.0),
    "cfloat spec+hits thr=1e9": dict(kind=capi.KIND_FLOAT_COMPLEX, flags=3, raw=x, thr=1e9),
    "int16 spec+hits": dict(kind=capi.KIND_SHORT_COMPLEX, flags=3, raw=raw16, thr=10.0),
    "int16 spec only": dict(kind=capi.KIND_SHORT_COMPLEX, flags=1, raw=raw16, thr=10.0),
    "int8 spec only": dict(kind=capi.KIND_BYTE_COMPLEX, flags=1, raw=raw8, thr=10.0),
}
import os
flt = sys.argv[1].split(",") if len(sys.argv) > 1 else None
if flt:
    variants = {k: v for k, v in variants.items() if any(f in k for f in flt)}
tag = os.path.basename(os.environ.get("SCN_LIB", "default")).replace("lib_", "").replace(".so", "")
plans = {}
for name, v in variants.items():
    plans[name] = Plan(n, 8000000, v["thr"], kind=v["kind"], enob=12 if v["kind"] != 1 else 8, max_batch=nb,
                       max_hits=nb * 64, flags=v["flags"])
res = {k: [] for k in variants}
K = 20
for rnd in range(5):
    for name, v in variants.items():
        p = plans[name]
        ext = torch.cuda.ExternalStream(p.stream_handle, device=dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for k in range(3):
            p.submit_device(0, v["raw"], nb, fc, sync_producer=False); p.wait(0); p.collect(0, False, False)
        torch.cuda.synchronize()
        e0.record(ext)
        pend = [False, False]
        for k in range(K):
            s = k & 1
            if pend[s]: p.collect(s, False, False)
            p.submit_device(s, v["raw"], nb, fc, sync_producer=False); pend[s] = True
        e1.record(ext)
        for s in (0, 1):
            if pend[s]: p.collect(s, False, False)
        torch.cuda.synchronize()
        res[name].append(e0.elapsed_time(e1) / K * 1e3)
for name, v in variants.items():
    r = sorted(res[name])
    bps = {4: 12, 3: 8, 1: 6}[v["kind"]] if v["flags"] & 1 else {4: 8, 3: 4, 1: 2}[v["kind"]]
    print(f"{tag:10s} {name:28s} median {r[len(r)//2]:8.2f} us  min {r[0]:8.2f} us   {nb*n/r[len(r)//2]/1e3:7.1f} Gsamples/s  "
          f"{nb*n*bps/r[len(r)//2]/1e6:6.2f} TB/s algorithmic")

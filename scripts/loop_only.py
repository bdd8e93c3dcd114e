"""The double-buffered counts-only step loop alone (for tracing):
   python scripts/loop_only.py [steps] [records 0/1] [n] [kind] [batch]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scanner_amd import Plan, capi, synth
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
records = len(sys.argv) > 2 and sys.argv[2] == "1"
n = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
kind_name = sys.argv[4] if len(sys.argv) > 4 else "cfloat"
nb = int(sys.argv[5]) if len(sys.argv) > 5 else 8192
kind = {"cfloat": capi.KIND_FLOAT_COMPLEX, "int16": capi.KIND_SHORT_COMPLEX, "int8": capi.KIND_BYTE_COMPLEX}[kind_name]
dev = torch.device("cuda", 0)
R = 4
raws = []
for r in range(R):
    x = synth.cfloat_batch_torch(n, nb, seed=2 + 1000 * r, device=dev)
    if kind_name == "int16":
        x = torch.clamp(torch.round(x * 2047.0), -2048, 2047).to(torch.int16).contiguous()
    elif kind_name == "int8":
        x = torch.clamp(torch.round(x * 127.0), -128, 127).to(torch.int8).contiguous()
    raws.append(x)
outs = [torch.empty((nb, n), dtype=torch.float32, device=dev) for r in range(R)]
fc = 3e6 + 6e6 * np.arange(nb)
torch.cuda.synchronize()
plan = Plan(n, 8000000, 10.0, kind=kind, enob=8 if kind_name == "int8" else 12, max_batch=nb, max_hits=nb * 64)
buf = np.zeros(nb * 64, capi.HIT_DTYPE) if records else None
pend = [False, False]
t_settle = time.perf_counter()
k0 = 0
while time.perf_counter() - t_settle < 0.6 or (k0 & 1):  # out of the idle power state first (bench.py does the same)
    s = k0 & 1
    if pend[s]:
        plan.collect(s, want_power=False, want_hits=records, hits_out=buf)
    plan.submit_device(s, raws[k0 % R], nb, fc, None, sync_producer=False, d_power_db=outs[k0 % R])
    pend[s] = True
    k0 += 1
for s in (0, 1):
    plan.collect(s, want_power=False, want_hits=records, hits_out=buf)
pend = [False, False]
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(steps):
    s = k & 1
    if pend[s]:
        plan.collect(s, want_power=False, want_hits=records, hits_out=buf)
    plan.submit_device(s, raws[k % R], nb, fc, None, sync_producer=False, d_power_db=outs[k % R])
    pend[s] = True
for s in (0, 1):
    plan.collect(s, want_power=False, want_hits=records, hits_out=buf)
torch.cuda.synchronize()
print((time.perf_counter() - t0) / steps * 1e6, "us/step")

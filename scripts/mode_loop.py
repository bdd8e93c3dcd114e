"""The pipelined double-buffered step loop for ONE output mode (for rocprofv3 / A-B timing):
   python scripts/mode_loop.py <n> <kind cfloat|int16|int8> <batch> <flags 1 spectrum | 3 spectrum+hits | 2 hits only> [steps] [threshold]
prints us per step (HIP events on the plan's stream, inputs rotated over 4 batches)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scanner_amd import Plan, capi, synth
n, kind_name, nb, flags = int(sys.argv[1]), sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 300
thr = float(sys.argv[6]) if len(sys.argv) > 6 else 10.0
kind = {"cfloat": capi.KIND_FLOAT_COMPLEX, "int16": capi.KIND_SHORT_COMPLEX, "int8": capi.KIND_BYTE_COMPLEX}[kind_name]
dev = torch.device("cuda", 0)
raws = []
for r in range(4):
    x = synth.cfloat_batch_torch(n, nb, seed=2 + 1000 * r, device=dev)
    if kind_name == "int16":
        x = torch.clamp(torch.round(x * 2047.0), -2048, 2047).to(torch.int16).contiguous()
    elif kind_name == "int8":
        x = torch.clamp(torch.round(x * 127.0), -128, 127).to(torch.int8).contiguous()
    raws.append(x)
fc = 3e6 + 6e6 * np.arange(nb)
p = Plan(n, 8000000, thr, kind=kind, enob=8 if kind_name == "int8" else 12, max_batch=nb, max_hits=nb * 64, flags=flags)
ext = torch.cuda.ExternalStream(p.stream_handle, device=dev)
def loop(K):
    pend = [False, False]
    for k in range(K):
        s = k & 1
        if pend[s]: p.collect(s, False, False)
        p.submit_device(s, raws[k % 4], nb, fc, sync_producer=False); pend[s] = True
    for s in (0, 1):
        if pend[s]: p.collect(s, False, False)
loop(200)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(ext); loop(steps); e1.record(ext); torch.cuda.synchronize()
print(f"{n} {kind_name} batch {nb} flags {flags}: {e0.elapsed_time(e1) / steps * 1e3:.1f} us per step")
p.close()

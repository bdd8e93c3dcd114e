"""us per launch (33.5 M samples) for every size x wire format x DC x output mode: a quick look for pathological corners.
Inputs rotated over 3 batches (not the full cache-defeating footprint of bench.py: relative numbers)."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from scanner_amd import Plan, capi, synth
dev = torch.device('cuda', 0)
names = {capi.KIND_FLOAT_COMPLEX: "cfloat", capi.KIND_SHORT_COMPLEX: "int16", capi.KIND_SHORT: "int16 planar", capi.KIND_BYTE_COMPLEX: "int8"}
print(f"{'n':>5s} {'format':>13s} {'dc':>3s} | spectrum only | spectrum+hits | hits only | time-domain   (us per launch of 33.5 M samples; 256 / 512 points: 32768 buffers = 8.4 / 16.8 M samples)")
sizes = [int(a) for a in sys.argv[1:]] or [1024, 2048, 4096, 8192, 16384]
for n in sizes:
    nb = min(8192 * 4096 // n, 8192 * 4)
    base = [synth.cfloat_batch_torch(n, nb, seed=5 + r, device=dev) for r in range(3)]
    fc = 3e6 + 6e6 * np.arange(nb)
    for kind in (capi.KIND_FLOAT_COMPLEX, capi.KIND_SHORT_COMPLEX, capi.KIND_SHORT, capi.KIND_BYTE_COMPLEX):
        if kind == capi.KIND_FLOAT_COMPLEX:
            xs = base
        elif kind == capi.KIND_BYTE_COMPLEX:
            xs = [torch.clamp(torch.round(x * 127.0), -128, 127).to(torch.int8).contiguous() for x in base]
        else:
            xs = [torch.clamp(torch.round(x * 2047.0), -2048, 2047).to(torch.int16) for x in base]
            if kind == capi.KIND_SHORT:   # planar: I[n] then Q[n]
                xs = [x.permute(0, 2, 1).contiguous() for x in xs]
            else:
                xs = [x.contiguous() for x in xs]
        for dc in ((False,) if kind == capi.KIND_FLOAT_COMPLEX else (False, True)):
            row = []
            for flags, mode in ((1, capi.MODE_FREQUENCY_DOMAIN), (3, capi.MODE_FREQUENCY_DOMAIN), (2, capi.MODE_FREQUENCY_DOMAIN), (3, capi.MODE_TIME_DOMAIN)):
                # threshold: bench.py's default (the same margin over the noise mean at every size)
                p = Plan(n, 8000000, 10.0 + 5.0 * np.log10(n / 4096.0), kind=kind, enob=8 if kind == capi.KIND_BYTE_COMPLEX else 12, correct_dc=dc, max_batch=nb,
                         max_hits=nb * 64, flags=flags, mode=mode)
                ext = torch.cuda.ExternalStream(p.stream_handle, device=dev)
                td = mode == capi.MODE_TIME_DOMAIN
                def coll(s):
                    if td: p.collect_time_domain(s)
                    else: p.collect(s, False, False)
                res = []
                import time
                t_settle = time.perf_counter()  # out of the idle power state first: creating a plan (allocations) lets the GPU fall
                k = 0                           # back, and the first ~100 launches after that run at up to half speed
                while time.perf_counter() - t_settle < 0.35:
                    p.submit_device(k & 1, xs[k % 3], nb, fc, sync_producer=False); coll(k & 1); k += 1
                for rnd in range(3):
                    pend = [False, False]
                    for k in range(40):
                        p.submit_device(k & 1, xs[k % 3], nb, fc, sync_producer=False); coll(k & 1)
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    K = 60
                    e0.record(ext)
                    for k in range(K):
                        s = k & 1
                        if pend[s]: coll(s)
                        p.submit_device(s, xs[k % 3], nb, fc, sync_producer=False); pend[s] = True
                    e1.record(ext)
                    for s in (0, 1):
                        if pend[s]: coll(s)
                    torch.cuda.synchronize()
                    res.append(e0.elapsed_time(e1) / K * 1e3)
                row.append(sorted(res)[1])
                p.close()
            print(f"{n:5d} {names[kind]:>13s} {str(dc)[0]:>3s} | {row[0]:13.1f} | {row[1]:13.1f} | {row[2]:9.1f} | {row[3]:11.1f}")

"""PCIe-inclusive throughput of the pinned double-buffered path (scn_host_buffer + scn_submit + scn_collect)."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from scanner_amd import Plan, capi, synth
for kind, name, n, nb in ((capi.KIND_FLOAT_COMPLEX, "cfloat", 4096, 4096), (capi.KIND_SHORT_COMPLEX, "int16", 8192, 2048),
                          (capi.KIND_BYTE_COMPLEX, "int8", 4096, 4096)):
    x = synth.quantize(synth.cfloat_batch(n, 64, seed=1), kind)
    p = Plan(n, 8000000, 10.0, kind=kind, enob=12 if kind != 1 else 8, max_batch=nb, max_hits=nb * 64, flags=capi.OUT_HITS)
    views = [p.host_buffer(s) for s in range(2)]
    raw = np.tile(x.view(np.uint8).reshape(-1), nb // 64)
    for v in views: v[:] = raw
    fc = np.zeros(nb)
    K = 30
    for warm in range(2):
        pend = [False, False]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k in range(K):
            s = k & 1
            if pend[s]: p.collect(s, False, False)
            p.submit(s, nb, fc); pend[s] = True
        for s in (0, 1):
            if pend[s]: p.collect(s, False, False)
        dt = time.perf_counter() - t0
    gb = K * nb * n * capi.BYTES_PER_SAMPLE[kind] / dt / 1e9
    print(f"{name:6s} n={n} batch={nb}: {K*nb*n/dt/1e9:6.2f} Gsamples/s  H2D {gb:5.1f} GB/s  ({dt/K*1e3:.2f} ms per batch)")
    p.close()

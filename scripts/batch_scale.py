"""Per-buffer kernel time vs batch size (ramp/tail share of a launch)."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from scanner_amd import Plan, capi, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device('cuda', 0)
big = 32768 * 4096 // n
x = synth.cfloat_batch_torch(n, big, seed=2, device=dev, max_tones=0)
for flags in (1, 3):
    for nb in (big // 32, big // 8, big // 4, big // 2, big):
        p = Plan(n, 8000000, 10.0, max_batch=nb, max_hits=nb * 64, flags=flags)
        ext = torch.cuda.ExternalStream(p.stream_handle, device=dev)
        fc = np.zeros(nb)
        ts = []
        for rnd in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            p.submit_device(0, x, nb, fc, sync_producer=False); p.wait(0); p.collect(0, False, False)
            torch.cuda.synchronize(); K = 10
            e0.record(ext)
            for k in range(K):
                p.submit_device(0, x, nb, fc, sync_producer=False); p.collect(0, False, False)
            e1.record(ext); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / K * 1e3)
        t = sorted(ts)[1]
        print(f"n={n} flags={flags} nb={nb:6d}  {t:8.2f} us/launch  {t/nb*1e3:7.3f} ns/buffer  {nb*n*12/t/1e6:5.2f} TB/s")
        p.close()

import sys, numpy as np, torch
sys.path.insert(0, '.')
from scanner_amd import Plan, capi, synth
from oracle import oracle as O
n, nb = 4096, 4
x = synth.cfloat_batch(n, nb, seed=2)
o = O.Oracle(n, 8000000, 1e9)
p_ref, _, _ = o.run(x)
for flags in (capi.OUT_SPECTRUM, capi.OUT_SPECTRUM | capi.OUT_HITS):
    with Plan(n, 8000000, 1e9, max_batch=nb, flags=flags) as plan:
        d = torch.from_numpy(x.view(np.uint8).reshape(-1)).cuda()
        plan.submit_device(0, d, nb)
        p, h, t = plan.collect(0)
    bad = np.abs(p - p_ref) > 1e-3
    print("flags", flags, "bad count per buffer", bad.sum(axis=1))
    b0 = np.flatnonzero(bad[0])
    print("bad r values:", sorted(set((b0 >> 8).tolist())))
    print("bad t values (first 40):", sorted(set((b0 & 255).tolist()))[:40], len(set((b0 & 255).tolist())))
    print("first bad k:", b0[:20])

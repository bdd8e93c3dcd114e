"""How close to the parity bar does the Welch path come on strong-tone streams?  (the 30-minute fuzz of round 6 read 9.97e-6 on a Welch case)
For K in (1, 2, 16): 24 streams of 4 PSDs each, delivery blocks from the synth recipe (noise sigma 0.05 + up to 4 tones of amplitude up to
0.5 per block), float and int16 samples; prints the largest value of the parity metric (tests/tolerances.py) per K and the worst case."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scanner_amd import WelchPlan, capi, synth
from oracle import oracle as O
from tests import tolerances as tol

N = 65536
for k in (1, 2, 16):
    worst = (0.0, None)
    over = 0
    with WelchPlan(N, k, max_psd=4) as w, WelchPlan(N, k, max_psd=4, kind=capi.KIND_SHORT_COMPLEX, enob=12) as wi:
        for seed in range(24):
            n_psd = 4
            blocks = n_psd * k + 1
            x = synth.cfloat_batch(N // 2, blocks, seed=1000 * k + seed, max_tones=4)
            for plan, kind in ((w, capi.KIND_FLOAT_COMPLEX), (wi, capi.KIND_SHORT_COMPLEX)):
                raw = synth.quantize(x, kind)
                flat = np.ascontiguousarray(raw).view(np.uint8).reshape(-1)
                plan.submit_device(0, torch.from_numpy(flat).cuda(), n_psd)
                got = plan.collect(0)
                ref = O.welch_raw(raw, kind, 12, False, N, k, n_psd)
                P_t, P_r = tol.db_to_power(np.where(np.isfinite(got), got, -300.0)), tol.db_to_power(np.where(np.isfinite(ref), ref, -300.0))
                m = (np.abs(P_t - P_r) / np.maximum(P_r, P_r.mean(axis=-1, keepdims=True))).max(axis=-1)
                pk = (P_r.max(axis=-1) / P_r.mean(axis=-1))
                j = int(np.argmax(m))
                if m[j] > worst[0]:
                    worst = (float(m[j]), (seed, kind, j, float(pk[j])))
                over += int((m > 1e-5).sum())
    print(f"K={k}: largest metric {worst[0]:.3g} (seed, kind, psd, peak/mean power: {worst[1]}), PSDs over the 1e-5 bar: {over} of {24 * 2 * 4}")

"""Accuracy tail of the large fused kernels on strong-tone buffers: max over bins of |dP| / max(P_bin, mean P) against the oracle, per
wire format, 1024 buffers each:   python scripts/acc16k.py [n ...]   (default 16384; round 5: 12000 14400 15000 16000, the all-float
mixed-radix kernels next to the 16384-point kernel whose last pass is in double for exactly this tail)"""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from scanner_amd import Plan, capi, synth
from oracle import oracle as O
from tests import tolerances as tol
nb = 256
for n in ([int(a) for a in sys.argv[1:]] or [16384]):
  for kind, enob, name in ((capi.KIND_BYTE_COMPLEX, 8, "int8"), (capi.KIND_SHORT_COMPLEX, 12, "int16"), (capi.KIND_FLOAT_COMPLEX, 12, "cfloat")):
      worst = []
      for seed in range(4):
          x = synth.cfloat_batch(n, nb, seed=900 + seed)
          raw = synth.quantize(x, kind)
          o = O.Oracle(n, 8000000, 1e9, kind=kind, enob=enob)
          p_ref, _, _ = o.run(raw, threads=8)
          with Plan(n, 8000000, 1e9, kind=kind, enob=enob, max_batch=nb) as plan:
              plan.submit_device(0, torch.from_numpy(np.ascontiguousarray(raw).view(np.uint8).reshape(-1)).cuda(), nb)
              p, _, _ = plan.collect(0)
          P = tol.db_to_power(np.where(np.isfinite(p_ref), p_ref, -300.0)); Pt = tol.db_to_power(np.where(np.isfinite(p), p, -300.0))
          rel = np.abs(Pt - P) / np.maximum(P, P.mean(axis=-1, keepdims=True))
          worst.append(rel.max(axis=-1))
      w = np.concatenate(worst)
      print(f"{n:6d} {name:6s}: max {w.max():.3g}  p99 {np.percentile(w, 99):.3g}  median {np.median(w):.3g}  buffers over 1e-5: {(w > 1e-5).sum()} of {len(w)}")

"""Per-phase shader-clock profile of the fused kernel (experiment build -DSCN_STAMPS=1).
   python scripts/build_variants.py stamps=SCN_STAMPS=1
   SCN_LIB=scanner_amd/variants/lib_stamps.so python scripts/stamp_profile.py [n] [flags]
Every workgroup's wave 0 accumulates cycles between phase boundaries over all its buffers; printed:
mean cycles per buffer per phase over the workgroups, and the share of the loop."""
import sys, os, numpy as np, torch
sys.path.insert(0, '.')
from scanner_amd import Plan, capi, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
flags = int(sys.argv[2]) if len(sys.argv) > 2 else 1
kindname = sys.argv[3] if len(sys.argv) > 3 else "cfloat"
kind = {"cfloat": capi.KIND_FLOAT_COMPLEX, "int16": capi.KIND_SHORT_COMPLEX}[kindname]
nb = 8192 * 4096 // n
dev = torch.device('cuda', 0)
R = 4
xs = [synth.cfloat_batch_torch(n, nb, seed=2 + r, device=dev) for r in range(R)]
if kindname == "int16":
    xs = [torch.clamp(torch.round(x * 2047.0), -2048, 2047).to(torch.int16).contiguous() for x in xs]
outs = [torch.empty((nb, n), dtype=torch.float32, device=dev) for r in range(R)]
fc = 3e6 + 6e6 * np.arange(nb)
p = Plan(n, 8000000, 10.0, kind=kind, enob=12, max_batch=nb, max_hits=nb * 64, flags=flags)
for k in range(12):   # pipelined like the bench; the last launch's stamps are read
    p.submit_device(k & 1, xs[k % R], nb, fc, sync_producer=False, d_power_db=outs[k % R])
    p.collect(k & 1, False, False)
torch.cuda.synchronize()
o = outs[(12 - 1) % R].cpu().numpy()
names = ["hit recording + loop", "wait for samples + convert/window", "issue next loads", "pass 1 + exch-1 writes",
         "barrier 1", "exch-1 reads + pass 2 + twiddle", "barrier 2", "exch-2 writes", "barrier 3",
         "exch-2 reads + pass 3 + dB + stores", "barrier 4"]
# workgroups = those whose first 12 floats look like stamps: count field == buffers per workgroup
cnt = o[:, 11]
grid = int((cnt > 0).sum()) if False else None
rows = []
for g in range(nb):
    c = o[g, 11]
    if c >= 1 and c == np.floor(c) and c <= nb and abs(o[g, :11].sum()) > 1000 and (o[g, :11] >= 0).all():
        rows.append(o[g, :12])
    else:
        break
rows = np.array(rows)
per = rows[:, :11] / rows[:, 11:12]
tot = per.sum(1).mean()
print(f"n={n} {kindname} flags={flags}: {len(rows)} workgroups, {rows[:,11].mean():.1f} buffers each, {tot:.0f} cycles per buffer per workgroup")
for i, nm in enumerate(names):
    print(f"  {nm:42s} {per[:, i].mean():8.0f} cyc  {100 * per[:, i].mean() / tot:5.1f} %   (p10 {np.percentile(per[:, i], 10):6.0f}  p90 {np.percentile(per[:, i], 90):6.0f})")

hc, hn = o[:len(rows), 16], o[:len(rows), 17]
print(f"wave 0 entered the hit path for {hn.sum() / rows[:, 11].sum() * 100:.1f} % of its buffers; {hc.sum() / max(hn.sum(), 1):.0f} cycles per entry "
      f"(p10 {np.percentile(hc / np.maximum(hn, 1), 10):.0f}  p90 {np.percentile(hc / np.maximum(hn, 1), 90):.0f})")
t0 = o[:len(rows), 12]; t1 = o[:len(rows), 13]; where = o[:len(rows), 14].astype(int); te = o[:len(rows), 15]
base = te.min()
st = (t0 - base) * 0.01; en = (t1 - base) * 0.01; ent = (te - base) * 0.01   # us
print(f"workgroup ENTRY times (us after the first): p50 {np.percentile(ent,50):.2f} p90 {np.percentile(ent,90):.2f} max {ent.max():.2f};  "
      f"prologue (entry -> first buffer): p10 {np.percentile(st-ent,10):.2f} p50 {np.percentile(st-ent,50):.2f} p90 {np.percentile(st-ent,90):.2f} max {(st-ent).max():.2f} us")
print(f"workgroup start times (us after the first): p50 {np.percentile(st,50):.2f} p90 {np.percentile(st,90):.2f} max {st.max():.2f};  "
      f"end: min {en.min():.2f} p50 {np.percentile(en,50):.2f} max {en.max():.2f};  duration p50 {np.percentile(en-st,50):.2f}")
late = st > 5.0
print(f"workgroups starting more than 5 us late: {int(late.sum())} of {len(rows)}")
import collections
percu = collections.Counter(where.tolist())
print("workgroups per CU histogram:", sorted(collections.Counter(percu.values()).items()))

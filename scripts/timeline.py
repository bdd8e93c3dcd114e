"""Prints the device timeline (kernels + memory copies) of the last few steps of a rocprofv3 --kernel-trace
--memory-copy-trace run:  python scripts/timeline.py <dir> [n_last_fft_launches]"""
import csv, glob, os, sys
d = sys.argv[1]
last = int(sys.argv[2]) if len(sys.argv) > 2 else 4
ev = []
for f in glob.glob(os.path.join(d, "**/*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"][:60], r.get("Queue_Id", "")))
for f in glob.glob(os.path.join(d, "**/*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", "")), ""))
ev.sort()
ffts = [i for i, e in enumerate(ev) if "scn_fft" in e[2]]
if not ffts:
    sys.exit("no fft kernels")
i0 = ffts[-min(len(ffts), last + 2)]
i1 = ffts[-2]
t0 = ev[i0][0]
for e in ev[i0:i1 + 1]:
    print(f"{(e[0]-t0)/1e3:9.2f} -> {(e[1]-t0)/1e3:9.2f} us  ({(e[1]-e[0])/1e3:7.2f})  q={e[3]:>3}  {e[2]}")

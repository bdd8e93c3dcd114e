#!/bin/bash
# Where does a sweep's time go with the gather in the loop?  The C4 steady-state leg under a kernel + copy trace: the device timeline of
# 500 us from the middle of the gather half of the leg, and per-kernel statistics.   bash scripts/r06_gather_probe.sh <tag> [centres]
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
TAG=${1:-probe}; C=${2:-2048}
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06_gather_probe/$TAG; mkdir -p $OUT
TR=$(mktemp -d /tmp/gtl.XXXXXX); export TR OUT
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $TR -- python3 bench.py --config c4 --centres $C --gather-every-sweep --steps 20 --warmup 5 --settle 0.05 \
   --no-cpu-baseline --no-overlap-leg --no-records-leg --no-hits-only-leg --no-copy-ref > $OUT/bench.json 2> $OUT/trace.log
python3 - <<'PY'
import csv, glob, os, json
import numpy as np
TR, OUT = os.environ['TR'], os.environ['OUT']
ev = []
for f in glob.glob(TR + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'K', r['Kernel_Name'][:70], r.get('Stream_Id', ''), r.get('Queue_Id', '')))
for f in glob.glob(TR + '/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'C', r.get('Direction', '') + ' ' + r.get('Bytes', r.get('Size', '')), r.get('Stream_Id', ''), ''))
ev.sort()
pk = [e for e in ev if 'scn_gather_pack' in e[3]]
with open(OUT + '/timeline.txt', 'w') as o:
    print('events', len(ev), 'pack kernels', len(pk), file=o)
    if pk:
        c = pk[len(pk) // 2][0] - 50_000
        for e in ev:
            if c <= e[0] < c + 500_000:
                print(f'{(e[0]-c)/1e3:9.1f} us  +{(e[1]-e[0])/1e3:8.1f} us  {e[2]} {e[3]:70s} s={e[4]} q={e[5]}', file=o)
        # the same length of time from the middle of the leg WITHOUT the gather: the FFT launches before the first pack kernel
        fft = [e for e in ev if e[2] == 'K' and 'scn_fft' in e[3] and e[0] < pk[0][0]]
        c = fft[-400][0] if len(fft) > 400 else fft[len(fft) // 2][0]
        print('--- without the gather', file=o)
        for e in ev:
            if c <= e[0] < c + 300_000:
                print(f'{(e[0]-c)/1e3:9.1f} us  +{(e[1]-e[0])/1e3:8.1f} us  {e[2]} {e[3]:70s} s={e[4]} q={e[5]}', file=o)
d = json.loads(open(OUT + '/bench.json').read())
print(json.dumps(d.get('gather_every_sweep'), indent=1)[:3000])
PY
head -120 $OUT/timeline.txt
rm -rf "$TR"

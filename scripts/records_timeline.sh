#!/bin/bash
# Timeline of the records leg (kernels + DMA) -> gpurun_out/records_tl/timeline.txt
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/records_tl; mkdir -p $OUT
TR=$(mktemp -d /tmp/rtl.XXXXXX)   # a directory of this run's own: traces of earlier runs on the box must not merge into the timeline
export TR
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $TR -- python3 bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-overlap-leg --no-hits-only-leg --no-copy-ref "$@" > $OUT/bench.json 2> $OUT/trace.log
python3 - <<'PY' > $OUT/timeline.txt
import csv, glob, os
ev = []
for f in glob.glob(os.environ['TR'] + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'K', r['Kernel_Name'][:60], r.get('Stream_Id', r.get('Queue_Id', ''))))
for f in glob.glob(os.environ['TR'] + '/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'C', r.get('Direction', '') + ' ' + r.get('Bytes', r.get('Size', '')), r.get('Stream_Id', '')))
ev.sort()
t0, t1 = ev[0][0], ev[-1][1]
print('events', len(ev), 'span ms', (t1 - t0) / 1e6)
# where do the list kernels live?  print three 700-us windows centred on compaction kernels found at 45 %, 60 %, 85 % of the span
names = sorted({e[3] for e in ev if e[2] == 'K'})
print('kernels:', names)
comp = [e[0] for e in ev if 'scn_hit_compact' in e[3]]
print('compaction kernels', len(comp))
for frac in (0.25, 0.75):   # the records leg, the zero-copy leg
    c = comp[int(len(comp) * frac)] - 20_000
    print(f'--- window at {frac:.2f}')
    for e in ev:
        if c <= e[0] < c + 700_000:
            print(f'{(e[0]-c)/1e3:9.1f} us  +{(e[1]-e[0])/1e3:8.1f} us  {e[2]} {e[3]:60s} q={e[4]}')
PY
head -c 600 $OUT/bench.json | tail -c 300; echo; wc -l $OUT/timeline.txt
rm -rf "$TR"

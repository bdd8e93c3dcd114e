#!/bin/bash
# Device timeline of bench.py's legs (kernels + copies) -> gpurun_out/records_tl/<tag>/{stats.txt,timeline.txt}
#   bash scripts/records_timeline.sh <tag> [bench.py arguments, e.g. --records-depth 4]
# stats.txt: the run cut into its legs (gaps > 2 ms between FFT launches), per leg the number of FFT launches, the mean
# launch-to-launch period, the FFT kernel's own duration (mean / p90 / max), and what ran beside it (list kernels, copies).
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
TAG=${1:-run}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/records_tl/$TAG; mkdir -p $OUT
TR=$(mktemp -d /tmp/rtl.XXXXXX)   # a directory of this run's own: traces of earlier runs on the box must not merge into the timeline
export TR OUT
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $TR -- python3 bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-overlap-leg --no-hits-only-leg --no-copy-ref "$@" > $OUT/bench.json 2> $OUT/trace.log
python3 - <<'PY'
import csv, glob, os
import numpy as np
TR, OUT = os.environ['TR'], os.environ['OUT']
ev = []
for f in glob.glob(TR + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'K', r['Kernel_Name'][:60], r.get('Stream_Id', r.get('Queue_Id', ''))))
for f in glob.glob(TR + '/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'C', r.get('Direction', '') + ' ' + r.get('Bytes', r.get('Size', '')), r.get('Stream_Id', '')))
ev.sort()
fft = [(e[0], e[1]) for e in ev if e[2] == 'K' and ('scn_fft' in e[3] or 'scn_big' in e[3])]
with open(OUT + '/stats.txt', 'w') as o:
    print('events', len(ev), 'fft launches', len(fft), file=o)
    legs, cur = [], [fft[0]]
    for a, b in zip(fft, fft[1:]):
        if b[0] - a[0] > 2_000_000:
            legs.append(cur); cur = []
        cur.append(b)
    legs.append(cur)
    for k, leg in enumerate(legs):
        if len(leg) < 50:
            continue
        s = np.array([x[0] for x in leg], float); d = np.array([x[1] - x[0] for x in leg], float) / 1e3
        per = np.diff(s) / 1e3
        t0, t1 = leg[0][0], leg[-1][1]
        side = {}
        for e in ev:
            if t0 <= e[0] <= t1 and not (e[2] == 'K' and ('scn_fft' in e[3] or 'scn_big' in e[3])):
                key = e[3].split('(')[-2][-28:] if e[2] == 'K' and '(' in e[3] else e[3][:28]
                side.setdefault(key, []).append((e[1] - e[0]) / 1e3)
        print(f'leg {k}: {len(leg)} launches, period mean {per.mean():.1f} p50 {np.median(per):.1f} us | fft dur mean {d.mean():.1f} p50 {np.median(d):.1f} p90 {np.quantile(d, .9):.1f} max {d.max():.1f} us | busy {d.sum() / (t1 - t0) * 1e3:.2f}', file=o)
        for key, v in sorted(side.items()):
            v = np.array(v)
            print(f'        beside: {key:30s} x{len(v):5d}  mean {v.mean():7.1f} us  p90 {np.quantile(v, .9):7.1f}', file=o)
with open(OUT + '/timeline.txt', 'w') as o:
    for k, leg in enumerate(legs):
        if len(leg) < 50:
            continue
        c = leg[len(leg) // 2][0] - 20_000
        print(f'--- leg {k}: 600 us from its middle', file=o)
        for e in ev:
            if c <= e[0] < c + 600_000:
                print(f'{(e[0]-c)/1e3:9.1f} us  +{(e[1]-e[0])/1e3:8.1f} us  {e[2]} {e[3]:60s} q={e[4]}', file=o)
PY
cat $OUT/stats.txt
rm -rf "$TR"

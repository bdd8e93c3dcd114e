#!/bin/bash
# The records loop of abi_bench on one time axis: its own per-call host timestamps (--host-log, steady_clock) merged with the
# device's kernels and copies (rocprofv3 --kernel-trace --memory-copy-trace: no API tracing, which slows every call down)
#   bash scripts/abi_trace.sh <tag> [abi_bench arguments, e.g. --mode view --depth 3]  -> gpurun_out/abi_trace/<tag>.txt
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
TAG=${1:-run}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/abi_trace; mkdir -p $OUT
TR=$(mktemp -d /tmp/abitr.XXXXXX)
export TR OUT TAG
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $TR -- scanner_amd/host/abi_bench --steps 600 --warmup 20 --host-log $TR/host.txt "$@" > $OUT/$TAG.json 2> $OUT/$TAG.log
python3 - <<'PY'
import csv, glob, os
TR, OUT, TAG = os.environ['TR'], os.environ['OUT'], os.environ['TAG']
ev = []
for f in glob.glob(TR + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'K', r['Kernel_Name'][:40] + ' ' + ' '.join(f"{k}={v}" for k, v in r.items() if k in ('Stream_Id', 'Queue_Id', 'Correlation_Id'))))
for f in glob.glob(TR + '/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'C', r.get('Direction', '')[:24] + ' ' + ' '.join(f"{k}={v}" for k, v in r.items() if k in ('Bytes', 'Size', 'Stream_Id', 'Queue_Id', 'Correlation_Id'))))
host = [l.split() for l in open(TR + '/host.txt')]
for w, s, t0, t1 in host:
    ev.append((int(float(t0)), int(float(t1)), 'H', {'S': 'submit', 'C': 'collect+view', 'c': 'scn_collect'}[w] + ' slot ' + s))
ev.sort()
fft = [e for e in ev if e[2] == 'K' and 'scn_fft' in e[3]]
hs = [e for e in ev if e[2] == 'H']
with open(f'{OUT}/{TAG}.txt', 'w') as o:
    print(f"host calls {len(hs)}, fft launches {len(fft)}; host clock vs device clock: first timed submit at {hs[0][0]}, fft launches span {fft[0][0]} .. {fft[-1][1]}", file=o)
    mid = hs[len(hs) // 2][0]
    for e in ev:
        if mid <= e[0] < mid + 800_000:
            print(f"{(e[0]-mid)/1e3:9.1f} us  +{(e[1]-e[0])/1e3:8.1f} us  {e[2]} {e[3]}", file=o)
PY
rm -rf $TR

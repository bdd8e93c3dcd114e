#!/bin/bash
# End-to-end rate of the kept class surface: SyntheticSource (replaying 64 generated buffers) -> SampleQueue (writing into the
# consumers' pinned slots: a ring per consumer thread) -> batched ProcessSamples worker(s) -> scn_submit -> scn_collect +
# scn_hits_view -> stdout (/dev/null).  Each configuration runs twice, N and 3N sweeps: the difference is the steady-state
# rate (plan creation, pinning the slots and the discarded warm-up sweep are in both runs).
cd "$GRAFT_REPO_ROOT/scanner_amd/host"
run() { ./scan_synth "$@" --replay 64 --sigma 0.05 --threshold 30 --start 0 > /dev/null 2> /tmp/err.txt; grep -E "^seconds|^buffers|^producer|^staging" /tmp/err.txt; }
pair() { n=$1; it=$2; shift 2
  a=$(run --n $n "$@" --niterations $it); b=$(run --n $n "$@" --niterations $((3 * it)))
  python3 - "$n" "$a" "$b" "$*" <<'PY'
import re, sys
n = int(sys.argv[1]); a, b, what = sys.argv[2], sys.argv[3], sys.argv[4]
f = lambda s: (int(re.search(r"buffers (\d+)", s).group(1)), float(re.search(r"seconds ([\d.]+)", s).group(1)))
(ba, ta), (bb, tb) = f(a), f(b)
print(f"== --n {n} {what}: {ba} buffers in {ta:.3f} s, {bb} in {tb:.3f} s -> steady state {(bb - ba) * n / (tb - ta) / 1e6:.0f} Msamples/s "
      f"({(tb - ta) / (bb - ba) * 1e6:.2f} us per buffer); whole run {bb * n / tb / 1e6:.0f}")
print("   " + [l for l in b.splitlines() if l.startswith("producer")][0])
st = [l for l in b.splitlines() if l.startswith("staging")][0]
print("   " + st)   # WHICH path ran: a silent fall-back to the copying worker is a different measurement
if "--threads 1" in what:
    assert st.startswith("staging: 1 of 1"), "one consumer must run the zero-copy path"
if "--threads 2" in what:
    assert st.startswith("staging: 2 of 2"), "both consumers must run the zero-copy path (a ring each, round 6)"
PY
}
pair 8192 150 --kind short_complex --enob 12 --stop 24576e6 --batch 2048 --depth 8192 --threads 1
pair 8192 150 --kind short_complex --enob 12 --stop 24576e6 --batch 2048 --depth 8192 --threads 2
pair 4096 75 --kind float --stop 49152e6 --batch 4096 --depth 16384 --threads 1
pair 4096 75 --kind float --stop 49152e6 --batch 4096 --depth 16384 --threads 2
pair 8192 150 --kind byte --enob 8 --stop 24576e6 --batch 2048 --depth 8192 --threads 1
pair 8192 150 --kind byte --enob 8 --stop 24576e6 --batch 2048 --depth 8192 --threads 2

#!/bin/bash
# End-to-end rate of the kept class surface: SyntheticSource (replaying 64 generated buffers) -> SampleQueue ->
# batched ProcessSamples worker(s) -> pinned slots -> scn_submit -> scn_collect with records -> stdout (/dev/null).
cd "$GRAFT_REPO_ROOT/scanner_amd/host"
run() { echo "== $*"; ./scan_synth "$@" --replay 64 --sigma 0.05 --threshold 30 --start 0 > /dev/null 2> /tmp/err.txt; tail -2 /tmp/err.txt; }
run --n 8192 --kind short_complex --enob 12 --stop 24576e6 --niterations 150 --batch 2048 --depth 8192 --threads 1
run --n 8192 --kind short_complex --enob 12 --stop 24576e6 --niterations 150 --batch 2048 --depth 8192 --threads 2
run --n 4096 --kind float --stop 49152e6 --niterations 75 --batch 4096 --depth 16384 --threads 1
run --n 4096 --kind float --stop 49152e6 --niterations 75 --batch 4096 --depth 16384 --threads 2
run --n 8192 --kind byte --enob 8 --stop 24576e6 --niterations 150 --batch 2048 --depth 8192 --threads 2

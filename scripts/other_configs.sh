#!/bin/bash
# The other BASELINE configs / wire formats, back to back on one box -> gpurun_out/configs_<tag>.jsonl
# usage: scripts/other_configs.sh <tag>
TAG=${1:-r03}
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/configs_$TAG.jsonl
: > $OUT
run() { python3 bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 >> $OUT; }
run --n 1024 --batch 32768
run --n 2048 --batch 16384
run --n 4096 --batch 8192
run --n 8192 --batch 4096
run --n 8192 --batch 8192
run --n 8192 --batch 4096 --kind int16
run --n 8192 --batch 8192 --kind int16
run --n 4096 --batch 8192 --kind int16
run --n 4096 --batch 8192 --kind int8
run --n 4096 --batch 2048
run --config c4
run --config c4 --centres 2048 --sweeps-per-launch 1   # the C4 per-GPU share at 8 GPUs, a launch per sweep (overlapped slots, three in flight)
run --n 8192 --batch 4096 --kind int8               # the reference's defaults: --count 8192 on a HackRF (int8)
run --n 16384 --batch 2048
run --n 16384 --batch 2048 --kind int16
run --n 512 --batch 65536                           # several buffers per workgroup (scn_fft_small_kernel)
run --n 256 --batch 131072
run --n 512 --batch 65536 --kind int16
run --n 65536 --batch 512 --steps 200 --warmup 20    # the four-step pair (scn_big.hip)
run --n 65536 --batch 512 --kind int16 --steps 200 --warmup 20
run --n 32768 --batch 1024 --steps 200 --warmup 20   # the four-step pair, 256 x 128
run --n 32768 --batch 1024 --kind int16 --steps 200 --warmup 20
run --n 128 --batch 262144                          # eight threads per buffer (scn_fft_tiny_kernel)
run --n 64 --batch 262144
run --n 128 --batch 262144 --per-buffer-centres      # as until round 4: 8 bytes of centre frequency per buffer with every submit (default now: a run of the plan's GPU-resident table)
run --n 16 --batch 524288                           # one thread per buffer
run --n 1000 --batch 32768                           # the mixed-radix fused kernels (scn_mixed.hip)
run --n 1000 --batch 32768 --kind int16
run --n 3000 --batch 11184
run --n 5000 --batch 6710
run --n 6000 --batch 5592
run --n 6000 --batch 5592 --kind int16
run --n 10000 --batch 3355
run --n 1536 --batch 21845                          # the 3 * 2^k / 5 * 2^k / 15 * 2^k families
run --n 3072 --batch 10922
run --n 5120 --batch 6553
run --n 7680 --batch 4369
run --n 10240 --batch 3276                          # (from 10240 up: two virtual threads per thread, pass 3 in double)
run --n 8192 --batch 4096 --kind int16 --plan-mode hits    # what ProcessSamples::ThreadWorker's plans run at the reference's defaults
run --n 8192 --batch 4096 --kind int8 --plan-mode hits
run --n 8192 --batch 4096 --kind int16 --time-domain       # the CLI's default mode (scan.cpp:87)
run --n 1023 --batch 4096 --steps 20 --warmup 3      # Bluestein (a size without a fused kernel)
run --n 12000 --batch 2796                           # beyond 10000: two virtual threads per thread, one in-place exchange
run --n 12000 --batch 2796 --kind int16
run --n 16000 --batch 2097
run --n 11000 --batch 3050 --steps 20 --warmup 3     # Bluestein: not 5-smooth
run --welch --welch-psd 32 --steps 100 --warmup 10
run --welch --welch-psd 8 --welch-pinned --steps 20 --warmup 3
python3 - <<PY
import json
for l in open("$OUT"):
    l = l.strip()
    if not l: continue
    d = json.loads(l); r = d.get("roofline") or {}; o = d.get("overlap") or {}
    w = d.get("with_hit_records") or {}
    print(f'{d["config"]["workload"][:74]:74s} {d["value"]/1e3:7.1f} Gs/s {d["ms_per_step"]*1e3:8.1f} us  frac {r.get("frac")}  overlap {o.get("value", 0)/1e3:7.1f} Gs/s  records {w.get("value", 0)/1e3:7.1f} Gs/s collect {w.get("collect_with_records_us")} us')
PY

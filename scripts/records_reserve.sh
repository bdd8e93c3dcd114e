#!/bin/bash
# the records legs (four submits in flight) with CUs left to the list kernels / the device-to-host blit: SCN_EXP_RESERVE_CUS
mkdir -p gpurun_out/oa
for r in 0 4 8 16 32 0; do
  SCN_EXP_RESERVE_CUS=$r python3 bench.py --steps 300 --no-cpu-baseline --no-overlap-leg --no-hits-only-leg --no-copy-ref 2>/dev/null | tail -1 > /tmp/rr.json
  python3 - <<PY
import json
d = json.loads(open('/tmp/rr.json').read()); w = d['with_hit_records']
print(f"reserve {$r:3d} CUs: contract leg {d['value']/1e3:6.1f} Gs/s | records (copy) {w['value']/1e3:6.1f}  two in flight {w['two_in_flight']['value']/1e3:6.1f}  view {w['zero_copy_view']['value']/1e3:6.1f}")
PY
done | tee gpurun_out/oa/records_reserve.txt

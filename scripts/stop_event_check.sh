# Does completing the per-launch event through the kernel's own dispatch packet (hipExtLaunchKernel's stopEvent) instead
# of a marker packet behind the kernel shorten the launch-to-launch time?  (us per step, counts-only double-buffered loop)
for shape in "4096 cfloat 8192" "4096 cfloat 16384" "4096 cfloat 4096" "4096 cfloat 2048" "1024 cfloat 32768" "2048 cfloat 16384" "4096 int16 8192" "4096 int16 4096" "2048 int16 16384"; do
  for v in 0 1 0 1; do echo -n "stop_event_in_packet=$v $shape: "; SCN_EXP_STOP_EVENT=$v python3 scripts/loop_only.py 2000 0 $shape 2>/dev/null | tail -1; done; done

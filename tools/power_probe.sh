#!/bin/bash
# power_probe.sh: socket power and shader clock while the headline bench runs (rocm-smi polled every 0.25 s)
( python bench.py --steps 4000 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/power_probe_bench.log 2>&1 ) &
bp=$!
sleep 4
for i in $(seq 1 12); do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|Temperature \(Sensor (junction|edge)" | tr '\n' ' ' | sed 's/GPU\[0\]//g; s/\s\+/ /g'
  echo
  sleep 0.25
done
wait $bp
tail -1 gpurun_out/power_probe_bench.log | cut -c1-200

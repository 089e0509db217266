#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 ) > gpurun_out/r3i_pytest.log 2>&1
tail -3 gpurun_out/r3i_pytest.log
timeout 300 python tools/per_epoch_breakdown.py > gpurun_out/r3i_breakdown.log 2>&1
NAV=tests/golden/20feb2022.rnx
: > gpurun_out/r3i_cli.log
for sink in /dev/null /dev/shm/cli.ishort; do for rep in 1 2 3; do
  t0=$(date +%s.%N)
  galileo-sdr-sim_amd/galileo-sdr-sim -e $NAV -l -6,51,100 -t 2022/02/20,12:00:00 -d 120 -P 0 -o $sink 2>&1 | grep -E "Process time" | tr '\n' ' ' >> gpurun_out/r3i_cli.log
  t1=$(date +%s.%N); echo " wall $(echo "$t1 - $t0" | bc) s -> $sink" >> gpurun_out/r3i_cli.log; rm -f /dev/shm/cli.ishort
done; done
for w in 0 4 8 16; do echo -n "writers $w: " >> gpurun_out/r3i_cli.log; galileo-sdr-sim_amd/galileo-sdr-sim -e $NAV -l -6,51,100 -t 2022/02/20,12:00:00 -d 120 -P 0 --writers $w -o /dev/shm/cli.ishort 2>&1 | grep "Process time" >> gpurun_out/r3i_cli.log; rm -f /dev/shm/cli.ishort; done
cat gpurun_out/r3i_breakdown.log gpurun_out/r3i_cli.log

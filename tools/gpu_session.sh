#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 ) > gpurun_out/r3h_pytest.log 2>&1
tail -3 gpurun_out/r3h_pytest.log
timeout 300 python tools/per_epoch_latency.py > gpurun_out/r3h_latency.log 2>&1
timeout 300 python tools/per_epoch_latency.py 1040 >> gpurun_out/r3h_latency.log 2>&1
NAV=tests/golden/20feb2022.rnx
for sink in /dev/null /dev/shm/cli.ishort; do for rep in 1 2 3; do
  /usr/bin/time -f "wall %e s" galileo-sdr-sim_amd/galileo-sdr-sim -e $NAV -l -6,51,100 -t 2022/02/20,12:00:00 -d 120 -P 0 -o $sink 2>&1 | grep -E "Process time|wall" | tr '\n' ' ' >> gpurun_out/r3h_cli.log; echo " -> $sink" >> gpurun_out/r3h_cli.log; rm -f /dev/shm/cli.ishort
done; done
GAL_SINK=stream galileo-sdr-sim_amd/galileo-sdr-sim -e $NAV -l -6,51,100 -t 2022/02/20,12:00:00 -d 120 -P 0 -o /dev/shm/cli.ishort 2>&1 | grep "Process time" >> gpurun_out/r3h_cli.log; rm -f /dev/shm/cli.ishort
for B in 32 64 128 256 512; do echo -n "batch $B: " >> gpurun_out/r3h_cli.log; galileo-sdr-sim_amd/galileo-sdr-sim -e $NAV -l -6,51,100 -t 2022/02/20,12:00:00 -d 120 -P 0 -B $B -o /dev/null 2>&1 | grep "Process time" >> gpurun_out/r3h_cli.log; done
( timeout 900 python tools/fuzz_parity.py 3000 81 ; timeout 900 python tools/fuzz_parity.py 150 82 big ; GAL_FUZZ_CBOC=1 timeout 900 python tools/fuzz_parity.py 1500 83; timeout 600 python tools/fuzz_scenarios.py 30 84 ) > gpurun_out/r3h_fuzz.log 2>&1
cat gpurun_out/r3h_latency.log gpurun_out/r3h_cli.log; grep -E "fuzz|scenario" gpurun_out/r3h_fuzz.log | tail -6

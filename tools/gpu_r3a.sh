#!/bin/bash
# round-3 first GPU session: tests, sink probes, CLI sink timing, profiles, bench
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r3a_pytest.log 2>&1
nproc > gpurun_out/r3a_sink_probe.log; cat /sys/kernel/mm/transparent_hugepage/shmem_enabled >> gpurun_out/r3a_sink_probe.log; df -h /dev/shm >> gpurun_out/r3a_sink_probe.log
timeout 300 tools/sink_probe /dev/shm/sp.bin 1189 1 2 4 8 16 32 >> gpurun_out/r3a_sink_probe.log 2>&1
for a in "0 0" "0 16" "128 16"; do timeout 120 tools/sink_probe_hip /dev/shm/sph.bin 1189 $a >> gpurun_out/r3a_sink_probe.log 2>&1; done
NAV=tests/golden/20feb2022.rnx
for w in 0 1 2 4 8 16 32; do
  for rep in 1 2; do
    echo "writers $w" >> gpurun_out/r3a_cli_sink.log
    galileo-sdr-sim_amd/galileo-sdr-sim -e $NAV -l -6,51,100 -t 2022/02/20,12:00:00 -d 120 -P 0 -o /dev/shm/cli.ishort --writers $w 2>&1 | grep "Process time" >> gpurun_out/r3a_cli_sink.log
    md5sum /dev/shm/cli.ishort >> gpurun_out/r3a_cli_sink.log; rm -f /dev/shm/cli.ishort
  done
done
echo "dev_null" >> gpurun_out/r3a_cli_sink.log
galileo-sdr-sim_amd/galileo-sdr-sim -e $NAV -l -6,51,100 -t 2022/02/20,12:00:00 -d 120 -P 0 -o /dev/null 2>&1 | grep "Process time" >> gpurun_out/r3a_cli_sink.log
echo "300 s, 16 writers" >> gpurun_out/r3a_cli_sink.log
galileo-sdr-sim_amd/galileo-sdr-sim -e $NAV -l -6,51,100 -t 2022/02/20,12:00:00 -d 300 -P 0 -o /dev/shm/cli.ishort 2>&1 | grep "Process time" >> gpurun_out/r3a_cli_sink.log
rm -f /dev/shm/cli.ishort
tools/profile_round.sh r03a > gpurun_out/r3a_profile.log 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r3a_bench_driver.json 2> gpurun_out/r3a_bench_driver.err
timeout 600 python bench.py --no-extras --no-cpu-baseline > gpurun_out/r3a_bench_100.json 2>> gpurun_out/r3a_bench_driver.err
timeout 600 python bench.py --no-extras --no-cpu-baseline --pipeline 1 > gpurun_out/r3a_bench_p1.json 2>> gpurun_out/r3a_bench_driver.err
tail -3 gpurun_out/r3a_pytest.log; cat gpurun_out/r3a_sink_probe.log gpurun_out/r3a_cli_sink.log

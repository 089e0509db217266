#!/bin/bash
# round-3 third GPU session: lean code walk, one leg per stitch thread
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r3c_pytest.log 2>&1
tail -3 gpurun_out/r3c_pytest.log
for p in 1 2; do timeout 300 python bench.py --no-extras --no-cpu-baseline --pipeline $p > gpurun_out/r3c_bench_p$p.json 2>gpurun_out/r3c_err.log; done
timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 20 > gpurun_out/r3c_bench_s20.json 2>>gpurun_out/r3c_err.log
timeout 300 python bench.py --no-extras --no-cpu-baseline --workload dyn --epochs 2999 --steps 20 > gpurun_out/r3c_bench_dyn.json 2>>gpurun_out/r3c_err.log
timeout 300 python bench.py --no-extras --no-cpu-baseline --channels 6 --pipeline 1 > gpurun_out/r3c_bench_ch6.json 2>>gpurun_out/r3c_err.log
GAL_SYNTH_LIB=galileo-sdr-sim_amd/variants/libgalsynth_w4.so timeout 300 python bench.py --no-extras --no-cpu-baseline --channels 6 --pipeline 1 > gpurun_out/r3c_bench_ch6_w4.json 2>>gpurun_out/r3c_err.log
GAL_SYNTH_LIB=galileo-sdr-sim_amd/variants/libgalsynth_w5.so timeout 300 python bench.py --no-extras --no-cpu-baseline --channels 6 --pipeline 1 > gpurun_out/r3c_bench_ch6_w5.json 2>>gpurun_out/r3c_err.log
GAL_SYNTH_LIB=galileo-sdr-sim_amd/variants/libgalsynth_w4.so timeout 300 python bench.py --no-extras --no-cpu-baseline --channels 4 --pipeline 1 > gpurun_out/r3c_bench_ch4_w4.json 2>>gpurun_out/r3c_err.log
timeout 300 python bench.py --no-extras --no-cpu-baseline --channels 4 --pipeline 1 > gpurun_out/r3c_bench_ch4.json 2>>gpurun_out/r3c_err.log
timeout 300 python tools/per_epoch_latency.py > gpurun_out/r3c_latency.log 2>&1
timeout 300 tools/trace_step.sh r3c > gpurun_out/r3c_trace.log 2>&1
timeout 300 tools/trace_step.sh r3c2 2 > gpurun_out/r3c_trace2.log 2>&1
( timeout 900 python tools/fuzz_parity.py 3000 41 ; timeout 900 python tools/fuzz_parity.py 150 42 big ; GAL_FUZZ_HOOKS=1 GAL_SCAN_SINGLE_LEGS=0 timeout 900 python tools/fuzz_parity.py 1500 43 ) > gpurun_out/r3c_fuzz.log 2>&1
python - <<'PY'
import json
for f in ("p1","p2","s20","dyn","ch6","ch6_w4","ch6_w5","ch4","ch4_w4"):
    try:
        d=json.loads(open("gpurun_out/r3c_bench_%s.json"%f).read().strip().splitlines()[-1]); r=d["roofline"]
        print(f, d["value"], d["ms_per_step"], "kernel", r["avg_kernel_ms"], "solo", r["standalone_kernel_ms"], "walk", r["avg_walk_ms"], d["config"]["walk_passes"])
    except Exception as e: print(f, "ERR", e)
PY
cat gpurun_out/r3c_latency.log; grep fuzz: gpurun_out/r3c_fuzz.log; tail -12 gpurun_out/r3c_trace.log; tail -30 gpurun_out/r3c_trace2.log

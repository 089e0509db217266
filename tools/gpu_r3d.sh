#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r3d_pytest.log 2>&1
tail -3 gpurun_out/r3d_pytest.log
for p in 1 2; do timeout 300 python bench.py --no-extras --no-cpu-baseline --pipeline $p > gpurun_out/r3d_bench_p$p.json 2>gpurun_out/r3d_err.log; done
timeout 300 python bench.py --no-extras --no-cpu-baseline --workload dyn --epochs 2999 --steps 20 > gpurun_out/r3d_bench_dyn.json 2>>gpurun_out/r3d_err.log
timeout 300 tools/trace_step.sh r3d > gpurun_out/r3d_trace.log 2>&1
timeout 300 tools/walk_silent_probe.sh > gpurun_out/r3d_silent.log 2>&1
( timeout 900 python tools/fuzz_parity.py 2000 51 ; timeout 900 python tools/fuzz_parity.py 100 52 big ; GAL_FUZZ_HOOKS=1 GAL_SCAN_SINGLE_LEGS=0 timeout 900 python tools/fuzz_parity.py 1500 53; GAL_FUZZ_HOOKS=1 GAL_SCAN_SINGLE_LEGS=0 timeout 900 python tools/fuzz_parity.py 60 54 big ) > gpurun_out/r3d_fuzz.log 2>&1
python - <<'PY'
import json
for f in ("p1","p2","dyn"):
    try:
        d=json.loads(open("gpurun_out/r3d_bench_%s.json"%f).read().strip().splitlines()[-1]); r=d["roofline"]
        print(f, d["value"], d["ms_per_step"], "kernel", r["avg_kernel_ms"], "solo", r["standalone_kernel_ms"], "walk", r["avg_walk_ms"], d["config"]["walk_passes"])
    except Exception as e: print(f, "ERR", e)
PY
grep fuzz: gpurun_out/r3d_fuzz.log; tail -12 gpurun_out/r3d_trace.log; cat gpurun_out/r3d_silent.log

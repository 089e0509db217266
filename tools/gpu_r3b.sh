#!/bin/bash
# round-3 second GPU session: walker-chain refactor (host SoA, in-stitch translation, ticketed publish, in-batch checkpoints)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r3b_pytest.log 2>&1
tail -3 gpurun_out/r3b_pytest.log
for p in 1 2; do timeout 300 python bench.py --no-extras --no-cpu-baseline --pipeline $p > gpurun_out/r3b_bench_p$p.json 2>gpurun_out/r3b_err.log; done
timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 20 > gpurun_out/r3b_bench_s20.json 2>>gpurun_out/r3b_err.log
timeout 300 python bench.py --no-extras --no-cpu-baseline --workload dyn --epochs 2999 --steps 20 > gpurun_out/r3b_bench_dyn.json 2>>gpurun_out/r3b_err.log
timeout 300 python tools/per_epoch_latency.py > gpurun_out/r3b_latency.log 2>&1
timeout 300 python tools/walk_stats.py > gpurun_out/r3b_walk_stats.log 2>&1
timeout 300 tools/trace_step.sh r3b > gpurun_out/r3b_trace.log 2>&1
( timeout 900 python tools/fuzz_parity.py 3000 31 ; timeout 900 python tools/fuzz_parity.py 150 32 big ; GAL_FUZZ_HOOKS=1 GAL_SCAN_SINGLE_LEGS=0 timeout 900 python tools/fuzz_parity.py 1500 33 ) > gpurun_out/r3b_fuzz.log 2>&1
python - <<'PY'
import json
for f in ("p1","p2","s20","dyn"):
    try:
        d=json.loads(open("gpurun_out/r3b_bench_%s.json"%f).read().strip().splitlines()[-1]); r=d["roofline"]
        print(f, d["value"], d["ms_per_step"], "kernel", r["avg_kernel_ms"], "solo", r["standalone_kernel_ms"], "walk", r["avg_walk_ms"], d["config"]["walk_passes"])
    except Exception as e: print(f, "ERR", e)
PY
cat gpurun_out/r3b_latency.log gpurun_out/r3b_walk_stats.log; tail -5 gpurun_out/r3b_fuzz.log; tail -40 gpurun_out/r3b_trace.log

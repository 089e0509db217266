#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_cboc.py -m gpu -x -q 2>&1 | tail -25 ) > gpurun_out/r3f_pytest_cboc.log 2>&1
tail -3 gpurun_out/r3f_pytest_cboc.log
timeout 300 python bench.py --no-extras --no-cpu-baseline --signal cboc --steps 30 > gpurun_out/r3f_bench_cboc.json 2>gpurun_out/r3f_err.log
GAL_BENCH_HOOKS=1 GAL_SYNTH_RW=0 timeout 300 python bench.py --no-extras --no-cpu-baseline --signal cboc --steps 30 > gpurun_out/r3f_bench_cboc_classic.json 2>>gpurun_out/r3f_err.log
timeout 300 python bench.py --no-extras --no-cpu-baseline --signal cboc --steps 30 --pipeline 1 > gpurun_out/r3f_bench_cboc_p1.json 2>>gpurun_out/r3f_err.log
( GAL_FUZZ_CBOC=1 timeout 900 python tools/fuzz_parity.py 3000 71 ; GAL_FUZZ_CBOC=1 timeout 900 python tools/fuzz_parity.py 120 72 big ) > gpurun_out/r3f_fuzz.log 2>&1
python - <<'PY'
import json
for f in ("cboc","cboc_classic","cboc_p1"):
    try:
        d=json.loads(open("gpurun_out/r3f_bench_%s.json"%f).read().strip().splitlines()[-1]); r=d["roofline"]
        print(f, d["value"], d["ms_per_step"], "kernel", r["avg_kernel_ms"], "solo", r["standalone_kernel_ms"], "mode", d["config"]["window_mode"], r["kernel"])
    except Exception as e: print(f, "ERR", e)
PY
grep fuzz: gpurun_out/r3f_fuzz.log

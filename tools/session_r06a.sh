#!/bin/bash
# round 6, session a: the new tests first, then the driver's bench command
set -u
tag=r06a
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests/test_parity_gpu.py tests/test_bench_contract.py tests/test_cli.py -m gpu -x -q -k "replay_check or stitch_records or plan_async or range_execute or epoch_ranges or small_batches or full_epoch or bench_json or time_overwrite" 2>&1 | tail -25 ) > gpurun_out/${tag}_pytest_new.log 2>&1
cat gpurun_out/${tag}_pytest_new.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench_default.json 2> gpurun_out/${tag}_bench.err
tail -5 gpurun_out/${tag}_bench.err
( GAL_PLAN_TIMING=1 timeout 300 python bench.py --steps 6 --warmup 2 --no-extras --no-cpu-baseline 2>&1 | grep "plan\]" | tail -12 ) > gpurun_out/${tag}_plan_timing.log
cat gpurun_out/${tag}_plan_timing.log
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r06a_bench_default.json").read().strip().splitlines()[-1]); r=d["roofline"]
print(d["value"], d["ms_per_step"], "frac", r["frac"], "kernel", r["avg_kernel_ms"], "overlapped", r["overlapped"]["avg_kernel_ms"])
print(json.dumps(d.get("configs",{}).get("fresh_plan")))
print(json.dumps(r.get("verify_sampled")))
print(json.dumps({k:(v if k!="fresh_plan" else "...") for k,v in d.get("configs",{}).items()}))
print(json.dumps(d.get("cpu_baseline")))
PY

#!/bin/bash
# round 6, last session: the driver's bench command with the round's PMC summaries in place (traffic / issue in the line), every recorded
# answer of the reference program replayed through the product CLI, the 4.092 MS/s leg on the exact-replay kernel for comparison
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r06z_bench_default.json 2> gpurun_out/r06z_bench_final.err
timeout 600 python bench.py --no-extras --no-cpu-baseline > gpurun_out/r06z_bench_100.json 2>> gpurun_out/r06z_bench_final.err
( timeout 900 python tools/ref_task_fuzz.py --replay tests/golden/ref_task_recorded.json --cli 0 0 6 2>&1 | tail -4 ) > gpurun_out/r06z_ref_task_replay_cli.log
( timeout 900 python tools/ref_task_fuzz.py --replay tests/golden/ref_task_recorded_long.json --cli 0 0 4 2>&1 | tail -3 ) > gpurun_out/r06z_ref_task_replay_cli_long.log
python - <<'PY' > gpurun_out/r06z_4092ksps_ab.log 2>&1
import json, os, sys
sys.path.insert(0, os.getcwd())
import torch
import bench
from __graft_entry__ import load_pkg
pkg = load_pkg()
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
for w in ("syn12_4092ksps", "syn12_4092ksps_exact", "syn12_4msps"):
    r = bench.leg_config(torch, pkg, w, 1199, 20, 0, streams)
    print(w, json.dumps(r))
PY
cat gpurun_out/r06z_4092ksps_ab.log | grep -v amdgpu; cat gpurun_out/r06z_ref_task_replay_cli.log | tail -2 | cut -c1-400; tail -1 gpurun_out/r06z_ref_task_replay_cli_long.log | cut -c1-400
python - <<'PY'
import json
for f in ("default","100"):
    d=json.loads(open("gpurun_out/r06z_bench_%s.json"%f).read().strip().splitlines()[-1]); r=d["roofline"]
    print(f, d["value"], d["ms_per_step"], "frac", r["frac"], "kernel", r["avg_kernel_ms"], "traffic", r["traffic"], "issue", r.get("issue",{}).get("frac"), "rocprof", r["rocprof_avg_kernel_ms"], r["frac_rocprof_standalone"], "fresh", d["configs"]["fresh_plan"]["ms_per_step"], d["configs"]["fresh_plan"]["ratio_to_resident_plan_step"])
PY

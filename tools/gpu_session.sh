#!/bin/bash
# Final measurements of a round on the GPU box: the GPU tests, the profiles to be judged (tools/profile_round.sh), the driver's bench
# command and its variants, the one-epoch latency, the rehearsal of the multi-GPU command.   tools/gpu_session.sh <tag>
set -u
tag=${1:-r06z}
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -8 ) > gpurun_out/${tag}_pytest.log 2>&1
tools/profile_round.sh $tag > gpurun_out/${tag}_profile.log 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench_default.json 2> gpurun_out/${tag}_bench.err
timeout 600 python bench.py --no-extras --no-cpu-baseline > gpurun_out/${tag}_bench_100.json 2>> gpurun_out/${tag}_bench.err
timeout 600 python bench.py --no-extras --no-cpu-baseline --no-fresh-plan --pipeline 1 > gpurun_out/${tag}_bench_p1.json 2>> gpurun_out/${tag}_bench.err
timeout 600 python bench.py --no-extras --no-cpu-baseline --no-fresh-plan --channels 9 > gpurun_out/${tag}_bench_ch9.json 2>> gpurun_out/${tag}_bench.err
timeout 300 python tools/per_epoch_latency.py > gpurun_out/${tag}_latency.log 2>&1
[ -x oracle/_ref/ref_task_hip ] && timeout 300 python tools/ref_task_goldens.py --hip > gpurun_out/${tag}_ref_task_hip_md5.log 2>&1
tail -3 gpurun_out/${tag}_pytest.log; tail -8 gpurun_out/${tag}_profile.log | cut -c1-300; cat gpurun_out/${tag}_latency.log; cut -c1-110 gpurun_out/${tag}_ref_task_hip_md5.log
python - $tag <<'PY'
import json,sys
tag=sys.argv[1]
for f in ("default","100","p1","ch9"):
    try:
        d=json.loads(open("gpurun_out/%s_bench_%s.json"%(tag,f)).read().strip().splitlines()[-1]); r=d["roofline"]
        print(f, d["value"], d["ms_per_step"], "frac", r["frac"], "kernel", r["avg_kernel_ms"], "sustained", r["sustained"]["frac"], "overlapped", r["overlapped"]["avg_kernel_ms"], "rocprof1", r["rocprof_avg_kernel_ms"], r["frac_rocprof_standalone"])
        if f=="default": print(json.dumps(d.get("configs")), json.dumps(d.get("e2e")), json.dumps(d.get("cpu_baseline")), json.dumps(r.get("verify_sampled")))
        if f=="100": print(json.dumps(d.get("configs",{}).get("fresh_plan")))
    except Exception as e: print(f,"ERR",e)
PY

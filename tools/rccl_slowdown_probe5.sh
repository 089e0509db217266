#!/bin/bash
# fifth round: process group created after the pre-roll (default now) and what of RCCL / c10d is left to cost 3 %
export HSA_ENABLE_IPC_MODE_LEGACY=0
fmt='import json,sys
for ln in sys.stdin:
    if ln.startswith("{"):
        d=json.loads(ln); r=d["roofline"]; print(sys.argv[1], "value", d["value"], "ms/step", d["ms_per_step"], "solo_kernel_ms", r["avg_kernel_ms"], "overlapped", r["overlapped"]["avg_kernel_ms"], "walk", r["avg_walk_ms"])'
args="--gpus 1 --steps 40 --warmup 5 --no-extras --no-cpu-baseline"
tr="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29521"
export GAL_BENCH_FORCE_DIST=1
for rep in 1 2; do
env -u GAL_BENCH_FORCE_DIST python bench.py $args 2>/dev/null | python -c "$fmt" "plain"
$tr bench.py $args 2>/dev/null | python -c "$fmt" "rccl_late(after preroll)"
GAL_BENCH_PG_ORDER=mid $tr bench.py $args 2>/dev/null | python -c "$fmt" "rccl_mid"
GAL_BENCH_PG_ORDER=early $tr bench.py $args 2>/dev/null | python -c "$fmt" "rccl_early"
TORCH_NCCL_ENABLE_MONITORING=0 TORCH_NCCL_ASYNC_ERROR_HANDLING=0 $tr bench.py $args 2>/dev/null | python -c "$fmt" "rccl_late no watchdog"
TORCH_NCCL_BLOCKING_WAIT=1 $tr bench.py $args 2>/dev/null | python -c "$fmt" "rccl_late blocking wait"
GAL_BENCH_BACKEND=gloo $tr bench.py $args 2>/dev/null | python -c "$fmt" "gloo_late"
done

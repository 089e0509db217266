#!/bin/bash
# second round: mechanism of the RCCL slowdown -- hardware-queue aliasing?  (GPU_MAX_HW_QUEUES, init order)
export HSA_ENABLE_IPC_MODE_LEGACY=0
fmt='import json,sys
for ln in sys.stdin:
    if ln.startswith("{"):
        d=json.loads(ln); r=d["roofline"]; print(sys.argv[1], "value", d["value"], "ms/step", d["ms_per_step"], "solo_kernel_ms", r["avg_kernel_ms"], "overlapped", r["overlapped"]["avg_kernel_ms"], "walk", r["avg_walk_ms"])'
args="--gpus 1 --steps 40 --warmup 5 --no-extras --no-cpu-baseline"
tr="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29521"
for rep in 1 2; do
GAL_BENCH_FORCE_DIST=1 $tr bench.py $args 2>/dev/null | python -c "$fmt" d_rccl_early
GAL_BENCH_FORCE_DIST=1 GAL_BENCH_PG_ORDER=late $tr bench.py $args 2>/dev/null | python -c "$fmt" e_rccl_late
GAL_BENCH_FORCE_DIST=1 GPU_MAX_HW_QUEUES=8 $tr bench.py $args 2>/dev/null | python -c "$fmt" f_rccl_early_q8
GAL_BENCH_FORCE_DIST=1 GPU_MAX_HW_QUEUES=16 $tr bench.py $args 2>/dev/null | python -c "$fmt" f_rccl_early_q16
GPU_MAX_HW_QUEUES=2 python bench.py $args 2>/dev/null | python -c "$fmt" g_plain_q2
GPU_MAX_HW_QUEUES=8 python bench.py $args 2>/dev/null | python -c "$fmt" g_plain_q8
GPU_MAX_HW_QUEUES=16 python bench.py $args 2>/dev/null | python -c "$fmt" g_plain_q16
done

#!/bin/bash
# third round: code walk enqueued on the walker stream (GAL_AUX_ON_WALK=1: one high-priority stream per handle in use)
export HSA_ENABLE_IPC_MODE_LEGACY=0
fmt='import json,sys
for ln in sys.stdin:
    if ln.startswith("{"):
        d=json.loads(ln); r=d["roofline"]; print(sys.argv[1], "value", d["value"], "ms/step", d["ms_per_step"], "solo_kernel_ms", r["avg_kernel_ms"], "overlapped", r["overlapped"]["avg_kernel_ms"], "walk", r["avg_walk_ms"])'
args="--gpus 1 --steps 40 --warmup 5 --no-extras --no-cpu-baseline"
tr="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29521"
for rep in 1 2; do
for aw in "" 1; do
export GAL_AUX_ON_WALK=$aw; [ -z "$aw" ] && unset GAL_AUX_ON_WALK
python bench.py $args 2>/dev/null | python -c "$fmt" "a_plain aw=$aw"
python bench.py $args --pipeline 1 2>/dev/null | python -c "$fmt" "a_plain_p1 aw=$aw"
GAL_BENCH_FORCE_DIST=1 $tr bench.py $args 2>/dev/null | python -c "$fmt" "d_rccl_early aw=$aw"
GAL_BENCH_FORCE_DIST=1 GAL_BENCH_PG_ORDER=late $tr bench.py $args 2>/dev/null | python -c "$fmt" "e_rccl_late aw=$aw"
GPU_MAX_HW_QUEUES=8 python bench.py $args 2>/dev/null | python -c "$fmt" "g_plain_q8 aw=$aw"
done
done

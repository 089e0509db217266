#!/bin/bash
# fourth round: does the slowdown need a k_synth grid larger than the resident block slots (768)?
export HSA_ENABLE_IPC_MODE_LEGACY=0
fmt='import json,sys
for ln in sys.stdin:
    if ln.startswith("{"):
        d=json.loads(ln); r=d["roofline"]; print(sys.argv[1], "value", d["value"], "ms/step", d["ms_per_step"], "solo_kernel_ms", r["avg_kernel_ms"], "overlapped", r["overlapped"]["avg_kernel_ms"], "walk", r["avg_walk_ms"])'
tr="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29521"
for ep in 384 768 1199 2400; do
args="--gpus 1 --steps 40 --warmup 5 --no-extras --no-cpu-baseline --epochs $ep"
python bench.py $args 2>/dev/null | python -c "$fmt" "plain      epochs=$ep"
GAL_BENCH_FORCE_DIST=1 $tr bench.py $args 2>/dev/null | python -c "$fmt" "rccl_early epochs=$ep"
python bench.py $args 2>/dev/null | python -c "$fmt" "plain      epochs=$ep"
GAL_BENCH_FORCE_DIST=1 $tr bench.py $args 2>/dev/null | python -c "$fmt" "rccl_early epochs=$ep"
done

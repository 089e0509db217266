#!/bin/bash
# Why does the same bench read lower under torch.distributed.run with the RCCL process group (world size 1)?
# a: plain python; b: torchrun, no process group; c: process group gloo; d: process group nccl (RCCL)
export HSA_ENABLE_IPC_MODE_LEGACY=0
fmt='import json,sys
for ln in sys.stdin:
    if ln.startswith("{"):
        d=json.loads(ln); r=d["roofline"]; print(sys.argv[1], "value", d["value"], "ms/step", d["ms_per_step"], "solo_kernel_ms", r["avg_kernel_ms"], "overlapped", r["overlapped"]["avg_kernel_ms"], "walk", r["avg_walk_ms"])'
args="--gpus 1 --steps 40 --warmup 5 --no-extras --no-cpu-baseline"
tr="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29521"
for rep in 1 2; do
python bench.py $args 2>/dev/null | python -c "$fmt" a_plain
$tr bench.py $args 2>/dev/null | python -c "$fmt" b_torchrun_nopg
OMP_NUM_THREADS=1 python bench.py $args 2>/dev/null | python -c "$fmt" b2_plain_omp1
GAL_BENCH_FORCE_DIST=1 GAL_BENCH_BACKEND=gloo $tr bench.py $args 2>/dev/null | python -c "$fmt" c_gloo
GAL_BENCH_FORCE_DIST=1 $tr bench.py $args 2>/dev/null | python -c "$fmt" d_rccl
done

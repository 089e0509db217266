#!/bin/bash
# seventh round: barriers through a gloo group, RCCL communicator lazy / eager
export HSA_ENABLE_IPC_MODE_LEGACY=0 GAL_BENCH_STEP_TIMES=1
tr="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29521"
f() { grep -E "^step times|ms_per_step" | sed -E 's/.*("value": [0-9.]*).*("ms_per_step": [0-9.]*).*/\1 \2/'; }
for steps in 40 20; do
args="--gpus 1 --steps $steps --warmup 5 --no-extras --no-cpu-baseline"
for rep in 1 2; do
echo "plain steps=$steps:";      python bench.py $args 2>&1 | f
echo "rccl lazy + gloo barrier steps=$steps:";  GAL_BENCH_FORCE_DIST=1 $tr bench.py $args 2>&1 | f
echo "rccl eager + gloo barrier steps=$steps:";  GAL_BENCH_FORCE_DIST=1 GAL_BENCH_RCCL=eager $tr bench.py $args 2>&1 | f
echo "rccl eager + rccl barrier steps=$steps:";  GAL_BENCH_FORCE_DIST=1 GAL_BENCH_RCCL=eager GAL_BENCH_BARRIER=rccl $tr bench.py $args 2>&1 | f
echo "rccl eager EARLY + gloo barrier steps=$steps:";  GAL_BENCH_FORCE_DIST=1 GAL_BENCH_RCCL=eager GAL_BENCH_PG_ORDER=early $tr bench.py $args 2>&1 | f
done
done

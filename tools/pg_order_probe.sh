#!/bin/bash
# The pipelined bench as a plain process and under torch.distributed.run with the RCCL process group created at different
# points (bench.py: GAL_BENCH_PG_ORDER / GAL_BENCH_RCCL / GAL_BENCH_BARRIER), world size 1.   -> gpurun_out/<tag>_pg_order.log
tag=${1:-rXX}
export HSA_ENABLE_IPC_MODE_LEGACY=0 GAL_BENCH_STEP_TIMES=1
tr="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29521"
f() { grep -E "^step times|ms_per_step" | sed -E 's/.*("value": [0-9.]*).*("ms_per_step": [0-9.]*).*/\1 \2/' | sed -E 's/, largest.*, end barrier/, end barrier/'; }
log=gpurun_out/${tag}_pg_order.log
: > $log
for steps in 20 100; do
args="--gpus 1 --steps $steps --warmup 5 --no-extras --no-cpu-baseline"
for rep in 1 2; do
{
echo "== plain process, steps=$steps";      python bench.py $args 2>&1 | f
echo "== default (group after the first step, RCCL lazy, gloo barriers), steps=$steps";  GAL_BENCH_FORCE_DIST=1 $tr bench.py $args 2>&1 | f
echo "== RCCL communicator eager after the first step, RCCL barriers, steps=$steps";  GAL_BENCH_FORCE_DIST=1 GAL_BENCH_RCCL=eager GAL_BENCH_BARRIER=rccl $tr bench.py $args 2>&1 | f
echo "== RCCL communicator eager after create() (mid), gloo barriers, steps=$steps";  GAL_BENCH_FORCE_DIST=1 GAL_BENCH_RCCL=eager GAL_BENCH_PG_ORDER=mid $tr bench.py $args 2>&1 | f
echo "== RCCL communicator eager FIRST (early), gloo barriers, steps=$steps";  GAL_BENCH_FORCE_DIST=1 GAL_BENCH_RCCL=eager GAL_BENCH_PG_ORDER=early $tr bench.py $args 2>&1 | f
echo "== the same with GPU_MAX_HW_QUEUES=8, steps=$steps";  GPU_MAX_HW_QUEUES=8 GAL_BENCH_FORCE_DIST=1 GAL_BENCH_RCCL=eager GAL_BENCH_PG_ORDER=early $tr bench.py $args 2>&1 | f
} >> $log
done
done
cat $log

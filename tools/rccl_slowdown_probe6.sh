#!/bin/bash
# sixth round: where inside the timed region does the RCCL run lose its 2-3 ms?  (GAL_BENCH_STEP_TIMES)
export HSA_ENABLE_IPC_MODE_LEGACY=0 GAL_BENCH_STEP_TIMES=1
args="--gpus 1 --steps 40 --warmup 5 --no-extras --no-cpu-baseline"
tr="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29521"
for rep in 1 2; do
echo "plain:";      python bench.py $args 2>&1 | grep -E "^step times|ms_per_step" | sed -E 's/.*("ms_per_step": [0-9.]*).*/\1/'
echo "rccl_late:";  GAL_BENCH_FORCE_DIST=1 $tr bench.py $args 2>&1 | grep -E "^step times|ms_per_step" | sed -E 's/.*("ms_per_step": [0-9.]*).*/\1/'
echo "rccl_early:"; GAL_BENCH_FORCE_DIST=1 GAL_BENCH_PG_ORDER=early $tr bench.py $args 2>&1 | grep -E "^step times|ms_per_step" | sed -E 's/.*("ms_per_step": [0-9.]*).*/\1/'
echo "gloo_late:";  GAL_BENCH_FORCE_DIST=1 GAL_BENCH_BACKEND=gloo $tr bench.py $args 2>&1 | grep -E "^step times|ms_per_step" | sed -E 's/.*("ms_per_step": [0-9.]*).*/\1/'
done

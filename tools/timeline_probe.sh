#!/bin/bash
# per-handle cycle (tools/step_timeline.py) of the pipelined bench: plain process vs RCCL process group (late / early)
tag=${1:-rXX}
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
args="--gpus 1 --steps 60 --warmup 5 --no-extras --no-cpu-baseline"
tr="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29521"
log=gpurun_out/${tag}_timeline.log
: > $log
run() {
    name=$1; shift
    rm -rf /tmp/tl_$name
    "$@" > /tmp/tl_$name.out 2>&1
    grep -o '"ms_per_step": [0-9.]*' /tmp/tl_$name.out | head -1 >> $log
    f=$(find /tmp/tl_$name -name '*kernel_trace.csv' | head -1)
    python3 tools/step_timeline.py "$f" $name >> $log
}
for rep in 1 2; do
run plain rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_plain -- python bench.py $args
GAL_BENCH_FORCE_DIST=1 run rccl_late rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_rccl_late -- $tr bench.py $args
GAL_BENCH_FORCE_DIST=1 GAL_BENCH_PG_ORDER=early GAL_BENCH_RCCL=eager run rccl_early rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_rccl_early -- $tr bench.py $args
GAL_BENCH_FORCE_DIST=1 GAL_BENCH_BACKEND=gloo run gloo_late rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_gloo_late -- $tr bench.py $args
done
cat $log

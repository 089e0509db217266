#!/bin/bash
# per-handle cycle of the pipelined bench (tools/step_timeline.py) with and without k_verify_carr (fault-injection build)
export TMPDIR=/tmp
args="--steps 60 --warmup 5 --no-extras --no-cpu-baseline"
run() {
    name=$1; shift
    rm -rf /tmp/tl_$name
    "$@" > /tmp/tl_$name.out 2>&1
    grep -o '"ms_per_step": [0-9.]*' /tmp/tl_$name.out | head -1
    f=$(find /tmp/tl_$name -name '*kernel_trace.csv' | head -1)
    python3 tools/step_timeline.py "$f" $name
}
GAL_BENCH_HOOKS=1 run verify rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_verify -- python bench.py $args
GAL_BENCH_HOOKS=1 GAL_G_NOVERIFY=1 run noverify rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_noverify -- python bench.py $args
for i in 1 2; do
GAL_BENCH_HOOKS=1 python bench.py $args --steps 100 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1
GAL_BENCH_HOOKS=1 GAL_G_NOVERIFY=1 python bench.py $args --steps 100 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1
done

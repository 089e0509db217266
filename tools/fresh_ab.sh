#!/bin/bash
# A/B of what a fresh-plan step pays beyond the resident-plan step (hooks build: GAL_WALK_PASSES forces the enqueued carrier passes)
export TMPDIR=/tmp
pr() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=d['configs']['fresh_plan']
print('$1', 'resident', d['ms_per_step'], 'walk', d['roofline']['avg_walk_ms'], 'kern', d['roofline']['overlapped']['avg_kernel_ms'], '| fresh3', f['ms_per_step'], 'walk', f['avg_walk_ms'], 'kern', f['avg_kernel_ms'], 'passes_max', f['walk_passes_max'], 'repeats', f['steps_with_a_repeated_synthesis'], '| fresh2', f['two_handles']['ms_per_step'])"; }
for rep in 1 2; do
GAL_BENCH_HOOKS=1 python bench.py --steps 40 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | pr "default        "
GAL_BENCH_HOOKS=1 GAL_WALK_PASSES=1 python bench.py --steps 40 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | pr "passes=1       "
GAL_BENCH_HOOKS=1 GAL_WALK_PASSES=2 python bench.py --steps 40 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | pr "passes=2       "
GAL_BENCH_HOOKS=1 GAL_WALK_PASSES=3 python bench.py --steps 40 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | pr "passes=3       "
done

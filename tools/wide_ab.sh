#!/bin/bash
# config 4 geometry (24 SVs, 25 MS/s): ONE wide k_synth_g launch (round 6) against two launches of 12, the second accumulating (rounds 3-5;
# hooks build, GAL_G_NARROW=1).  tools/wide_ab.sh [epochs] [steps]
export TMPDIR=/tmp
E=${1:-600}; K=${2:-8}
pr() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1', d['value'], 'Msamples/s', d['ms_per_step'], 'ms/step; kernel alone', r['avg_kernel_ms'], 'ms, frac', r['frac'], '; overlapped', r['overlapped']['avg_kernel_ms'], r['kernel'])"; }
for rep in 1 2; do
GAL_BENCH_HOOKS=1 python bench.py --workload syn24 --epochs $E --steps $K --warmup 2 --no-extras --no-cpu-baseline --no-fresh-plan 2>/dev/null | pr "wide  (1 launch of 24)  "
GAL_BENCH_HOOKS=1 GAL_G_NARROW=1 python bench.py --workload syn24 --epochs $E --steps $K --warmup 2 --no-extras --no-cpu-baseline --no-fresh-plan 2>/dev/null | pr "narrow (12 + 12 accum.) "
done

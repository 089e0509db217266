export TMPDIR=/tmp
pr() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1', d['value'], 'Msamples/s', d['ms_per_step'], 'ms/step; kernel alone', r['avg_kernel_ms'], 'ms')"; }
for bpe in 1 2 4 8; do
GAL_BENCH_HOOKS=1 GAL_G_BPE=$bpe python bench.py --workload syn24 --epochs 600 --steps 8 --warmup 2 --no-extras --no-cpu-baseline --no-fresh-plan 2>/dev/null | pr "wide bpe=$bpe E=600 "
done
GAL_BENCH_HOOKS=1 GAL_G_NARROW=1 python bench.py --workload syn24 --epochs 600 --steps 8 --warmup 2 --no-extras --no-cpu-baseline --no-fresh-plan 2>/dev/null | pr "narrow E=600     "
GAL_BENCH_HOOKS=1 python bench.py --workload syn24 --epochs 5999 --steps 3 --warmup 1 --no-extras --no-cpu-baseline --no-fresh-plan 2>/dev/null | pr "wide   E=5999 "
GAL_BENCH_HOOKS=1 GAL_G_NARROW=1 python bench.py --workload syn24 --epochs 5999 --steps 3 --warmup 1 --no-extras --no-cpu-baseline --no-fresh-plan 2>/dev/null | pr "narrow E=5999 "
GAL_BENCH_HOOKS=1 GAL_G_THREADS=512 python bench.py --workload syn24 --epochs 600 --steps 8 --warmup 2 --no-extras --no-cpu-baseline --no-fresh-plan 2>/dev/null | pr "wide (thr env 512: no effect on wide) "

#!/bin/bash
# quick bench summary: ms/step, standalone kernel ms, avg kernel ms, walk ms, checksum
for args in "$@"; do
python bench.py --no-cpu-baseline --no-extras $args 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$args', '| step', d['ms_per_step'], 'standalone', r['avg_kernel_ms'], 'overlapped', r['overlapped']['avg_kernel_ms'], 'walk', r.get('avg_walk_ms'), d['config']['output_checksum'])"
done

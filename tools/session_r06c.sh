#!/bin/bash
set -u
tag=r06c
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 3000 python -m pytest tests -m gpu -q --durations=12 2>&1 | tail -40 ) > gpurun_out/${tag}_pytest.log 2>&1
tail -25 gpurun_out/${tag}_pytest.log
GAL_BENCH_DEVICE=0 GAL_SCALE_STEPS=20 timeout 1500 tools/scale_node.sh ${tag}_rehearsal 2>&1 | tail -40

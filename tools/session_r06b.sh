#!/bin/bash
# round 6, session b: the whole GPU suite, then the profiles of the round
set -u
tag=r06b
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 3000 python -m pytest tests -m gpu -q -x --durations=15 2>&1 | tail -40 ) > gpurun_out/${tag}_pytest.log 2>&1
tail -30 gpurun_out/${tag}_pytest.log
tools/profile_round.sh $tag > gpurun_out/${tag}_profile.log 2>&1
tail -12 gpurun_out/${tag}_profile.log | cut -c1-300

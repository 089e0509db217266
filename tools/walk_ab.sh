#!/bin/bash
# same-box A/B of the walker chain's knobs (GAL_TEST_HOOKS build): verification off, carrier legs per epoch 4 / 8 / 16 / 32
fmt='import json,sys
for ln in sys.stdin:
    if ln.startswith("{"):
        d=json.loads(ln); r=d["roofline"]; print(sys.argv[1], "value", d["value"], "ms/step", d["ms_per_step"], "solo_kernel_ms", r["avg_kernel_ms"], "walk", r["avg_walk_ms"], "chk", d["config"]["output_checksum"])'
export GAL_BENCH_HOOKS=1
for i in 1 2; do
  for cfg in "X=0" "GAL_G_NOVERIFY=1" "GAL_WALK_LEGS=16" "GAL_WALK_LEGS=4" "GAL_WALK_LEGS=32" "GAL_WALK_LEGS=16 GAL_G_NOVERIFY=1"; do
    env $cfg python bench.py --no-extras --no-cpu-baseline "$@" 2>/dev/null | python -c "$fmt" "p2 $cfg"
  done
done
for cfg in "X=0" "GAL_G_NOVERIFY=1" "GAL_WALK_LEGS=16" "GAL_WALK_LEGS=4" "GAL_WALK_LEGS=32"; do
  env $cfg python bench.py --no-extras --no-cpu-baseline --pipeline 1 "$@" 2>/dev/null | python -c "$fmt" "p1 $cfg"
done

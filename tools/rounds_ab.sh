#!/bin/bash
# same-box A/B of k_synth_g's block layout (GAL_TEST_HOOKS build): one block per epoch (rounds 1-4: GAL_G_BPE=1) against contiguous
# chunk ranges in 1 / 2 / 3 / 4 rounds of the resident slots (GAL_G_ROUNDS), pipelined (2 handles) and one handle
fmt='import json,sys
for ln in sys.stdin:
    if ln.startswith("{"):
        d=json.loads(ln); r=d["roofline"]; print(sys.argv[1], "value", d["value"], "ms/step", d["ms_per_step"], "solo_kernel_ms", r["avg_kernel_ms"], "walk", r["avg_walk_ms"], "chk", d["config"]["output_checksum"])'
export GAL_BENCH_HOOKS=1
for i in 1 2; do
  for cfg in "GAL_G_BPE=1" "GAL_G_ROUNDS=1" "GAL_G_ROUNDS=2" "GAL_G_ROUNDS=3" "GAL_G_ROUNDS=4" "GAL_G_ROUNDS=6"; do
    env $cfg python bench.py --no-extras --no-cpu-baseline "$@" 2>/dev/null | python -c "$fmt" "p2 $cfg"
  done
done
for cfg in "GAL_G_BPE=1" "GAL_G_ROUNDS=1" "GAL_G_ROUNDS=2" "GAL_G_ROUNDS=3" "GAL_G_ROUNDS=4" "GAL_G_ROUNDS=6"; do
  env $cfg python bench.py --no-extras --no-cpu-baseline --pipeline 1 "$@" 2>/dev/null | python -c "$fmt" "p1 $cfg"
done

#!/bin/bash
# same-box A/B of k_verify_carr's share (GAL_TEST_HOOKS build): none, every leg, every 4th / 8th / 16th leg position per batch
fmt='import json,sys
for ln in sys.stdin:
    if ln.startswith("{"):
        d=json.loads(ln); r=d["roofline"]; print(sys.argv[1], "value", d["value"], "ms/step", d["ms_per_step"], "solo_kernel_ms", r["avg_kernel_ms"], "walk", r["avg_walk_ms"], "chk", d["config"]["output_checksum"])'
export GAL_BENCH_HOOKS=1
for i in 1 2; do
  for cfg in "GAL_G_NOVERIFY=1" "GAL_VERIFY_MOD=1" "GAL_VERIFY_MOD=4" "GAL_VERIFY_MOD=8" "GAL_VERIFY_MOD=16"; do
    env $cfg python bench.py --no-extras --no-cpu-baseline "$@" 2>/dev/null | python -c "$fmt" "p2 $cfg"
  done
done
for i in 1 2; do
for cfg in "GAL_G_NOVERIFY=1" "GAL_VERIFY_MOD=1" "GAL_VERIFY_MOD=4" "GAL_VERIFY_MOD=8" "GAL_VERIFY_MOD=16"; do
  env $cfg python bench.py --no-extras --no-cpu-baseline --pipeline 1 "$@" 2>/dev/null | python -c "$fmt" "p1 $cfg"
done
done

#!/bin/bash
# same-box A/B of a variant library (tools/build_variant.sh): tools/ab_lib.sh <variant.so> [alternations] -- bench args...
lib=$1; n=${2:-3}; shift 2; [ "$1" = "--" ] && shift
fmt='import json,sys
for ln in sys.stdin:
    if ln.startswith("{"):
        d=json.loads(ln); r=d["roofline"]; print(sys.argv[1], "value", d["value"], "ms/step", d["ms_per_step"], "solo_kernel_ms", r["avg_kernel_ms"], "checksum", d["config"]["output_checksum"])'
for i in $(seq $n); do
  python bench.py --no-extras --no-cpu-baseline "$@" 2>/dev/null | python -c "$fmt" "A(product)"
  GAL_SYNTH_LIB=$lib python bench.py --no-extras --no-cpu-baseline "$@" 2>/dev/null | python -c "$fmt" "B(variant)"
done

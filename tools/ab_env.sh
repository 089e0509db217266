#!/bin/bash
# same-box A/B of one environment switch: tools/ab_env.sh VAR [alternations] -- bench args...   (A = VAR unset, B = VAR=1)
var=$1; n=${2:-3}; shift 2; [ "$1" = "--" ] && shift
fmt='import json,sys
for ln in sys.stdin:
    if ln.startswith("{"):
        d=json.loads(ln); r=d["roofline"]; print(sys.argv[1], "value", d["value"], "ms/step", d["ms_per_step"], "solo_kernel_ms", r["avg_kernel_ms"], "walk", r["avg_walk_ms"])'
for i in $(seq $n); do
  env -u $var python bench.py --no-extras --no-cpu-baseline "$@" 2>/dev/null | python -c "$fmt" "A($var unset)"
  env $var=1 python bench.py --no-extras --no-cpu-baseline "$@" 2>/dev/null | python -c "$fmt" "B($var=1)    "
done

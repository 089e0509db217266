#!/bin/bash
# BASELINE config 5's per-rank units, one after the other on ONE GPU: every site of shard.LOCATIONS for 300 s through the host
# front-end and the engine (bench.py --workload locations --site k) -- SV count and ms per 300 s scenario per site: the 1-GPU
# baseline an 8-GPU run of config 5 is to be read against the day a node exists.  No scaling claim.
for k in 0 1 2 3 4 5 6 7; do
  python bench.py --workload locations --site $k --steps 20 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | python -c '
import json,sys
for ln in sys.stdin:
    if ln.startswith("{"):
        d=json.loads(ln); c=d["config"]
        print("site %s  SVs(max) %2d  epochs 2999  ms per 300 s scenario %.3f  = %.1f G samples/s  kernel alone %.3f ms  walker chain %.3f ms  passes %s  synth runs %s  family %s  checksum %s  %s" % (
            sys.argv[1], c["channels"], d["ms_per_step"], d["value"]/1e3, d["roofline"]["avg_kernel_ms"], d["roofline"]["avg_walk_ms"], c["walk_passes"], c["synth_runs_max"],
            c["kernel_family"], c["output_checksum"], c["workload"][44:90]))' $k
done

#!/bin/bash
# shard.epoch_range's balance read off ONE GPU: every rank of a world of 2 and of 4 does its share of the
# strong split (--shard scenario: its weighted epoch range of ONE 120 s scenario, prefix walked silently) ALONE on the device, one
# after the other, one handle; rank_imbalance = slowest rank's walker chain + synthesis over the mean.  Says nothing about scaling.
for world in 2 4 8; do
  python - $world <<'PY'
import json, subprocess, sys
world = int(sys.argv[1])
rows = []
for r in range(world):
    out = subprocess.run([sys.executable, "bench.py", "--shard", "scenario", "--as-rank", "%d/%d" % (r, world), "--pipeline", "1", "--steps", "20",
                          "--warmup", "3", "--no-extras", "--no-cpu-baseline"], capture_output=True, text=True).stdout
    d = [json.loads(l) for l in out.splitlines() if l.startswith("{")][-1]
    a = d["as_rank"]
    rows.append(a)
    print("world %d rank %d epochs %s  walk %.4f ms  kernel %.4f ms  step %.4f ms  legs walked %d" % (world, r, a["epochs"], a["avg_walk_ms"], a["avg_kernel_ms"], d["ms_per_step"], a["legs_walked"]))
t = [a["avg_walk_ms"] + a["avg_kernel_ms"] for a in rows]
print("world %d rank_imbalance (walk + kernel, each rank ALONE) %.3f   [shard.WALK_COST model target 1.00]" % (world, max(t) / (sum(t) / len(t))))
PY
done

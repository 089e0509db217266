#!/bin/bash
# HBM write/read traffic of k_synth only (two PMC passes): tools/pmc_wr.sh <tag>
export TMPDIR=/tmp
out=gpurun_out/pmcwr_$1
mkdir -p $out
cmd="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --pipeline 1"
timeout 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/p1 -- $cmd > $out/p1.log 2>&1
timeout 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/p2 -- $cmd > $out/p2.log 2>&1
python3 - $out <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/p*/*/*counter_collection.csv"):
    per = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        if "k_synth" not in r["Kernel_Name"]: continue
        per[r["Dispatch_Id"]][r["Counter_Name"]] = per[r["Dispatch_Id"]].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    for d in per.values():
        for k, v in d.items(): acc[k].append(v)
for k, v in sorted(acc.items()): print(k, round(sum(v)/len(v), 1), "KiB", len(v))
PY

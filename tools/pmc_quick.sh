#!/bin/bash
# quick single PMC pass: instruction counts of k_synth
export TMPDIR=/tmp
tag=$1
out=gpurun_out/pmcq_$tag
mkdir -p $out
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $out/p1 -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --pipeline 1 > $out/p1.log 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $out/p2 -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --pipeline 1 > $out/p2.log 2>&1
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
for k, v in sorted(acc.items()): print(k, sum(v)/len(v)/1e6, "M", len(v))
PY

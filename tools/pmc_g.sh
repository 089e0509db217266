#!/bin/bash
# pmc_g.sh <tag>: kernel trace + a few PMC passes of the default bench step (one handle), per kernel
export TMPDIR=/tmp
tag=$1
out=gpurun_out/pmcg_$tag
mkdir -p $out
cd /tmp
R=$GRAFT_REPO_ROOT
B="python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --pipeline 1"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/kt -- $B > $R/$out/kt.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $R/$out/p1 -- $B > $R/$out/p1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $R/$out/p2 -- $B > $R/$out/p2.log 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $R/$out/p3 -- $B > $R/$out/p3.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM --output-format csv -d $R/$out/p4 -- $B > $R/$out/p4.log 2>&1
cd $R
python3 - $out <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/p*/*/*counter_collection.csv"):
    per = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].split("<")[0]
        if "k_synth" not in k and "k_repair" not in k and "k_verify" not in k: continue
        per[(k, r["Dispatch_Id"])][r["Counter_Name"]] = per[(k, r["Dispatch_Id"])].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    for (k, _), d in per.items():
        for c, v in d.items(): acc[k][c].append(v)
for k in sorted(acc):
    for c, v in sorted(acc[k].items()): print("%-14s %-22s %10.3f M  (%d launches)" % (k, c, sum(v)/len(v)/1e6, len(v)))
for f in glob.glob(sys.argv[1] + "/kt/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        print("%-60s calls %5s avg %10.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3))
PY

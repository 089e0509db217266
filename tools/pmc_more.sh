#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/pmcm_$1
mkdir -p $out
rocprofv3 -L 2>/dev/null | grep -o "SQC\?_[A-Z_0-9]*" | sort -u | tr '\n' ' ' | head -c 6000 > $out/counters.txt
i=0
for ctrs in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH" "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_LDS" "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $ctrs --output-format csv -d $out/p$i -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --pipeline 1 > $out/p$i.log 2>&1 || echo "pass $i failed: $ctrs"
done
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
for k, v in sorted(acc.items()): print(k, round(sum(v)/len(v)/1e6, 3), "M")
PY
cat $out/counters.txt | head -c 3000

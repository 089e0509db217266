#!/bin/bash
# SQ_INSTS_VALU / SQ_WAVES of EVERY kernel of one step (not only k_synth): where the walker chain's instructions are.
#   tools/pmc_all_kernels.sh <tag>  -> gpurun_out/<tag>_pmc_all_kernels.log
tag=${1:-rXX}
export TMPDIR=/tmp
out=/tmp/pmcall_$tag
rm -rf $out; mkdir -p $out gpurun_out
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU --output-format csv -d $out/p1 -- \
    python bench.py --steps 3 --warmup 1 --preroll-ms 0 --no-cpu-baseline --no-extras --pipeline 1 > $out/p1.log 2>&1
python3 - $out > gpurun_out/${tag}_pmc_all_kernels.log <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/p*/*/*counter_collection.csv"):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    name = {}
    for r in csv.DictReader(open(f)):
        per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
        name[r["Dispatch_Id"]] = r["Kernel_Name"].split("(")[0].replace("void ", "")[:36]
    for d, c in per.items():
        for k, v in c.items():
            acc[name[d]][k].append(v)
tot = 0.0
for n in sorted(acc, key=lambda n: -sum(acc[n]["SQ_INSTS_VALU"]) / max(1, len(acc[n]["SQ_INSTS_VALU"]))):
    c = acc[n]
    avg = {k: sum(v) / len(v) for k, v in c.items()}
    if n.startswith("k_") : tot += avg.get("SQ_INSTS_VALU", 0.0) if not n.startswith("k_synth") else 0.0
    print("%-38s launches %4d  VALU %10.3f M  SALU %9.3f M  waves %8.0f  VALU/wave %9.0f" % (
        n, len(c["SQ_INSTS_VALU"]), avg.get("SQ_INSTS_VALU", 0) / 1e6, avg.get("SQ_INSTS_SALU", 0) / 1e6, avg.get("SQ_WAVES", 0),
        avg.get("SQ_INSTS_VALU", 0) / max(1.0, avg.get("SQ_WAVES", 1))))
print("walker chain + k_publish, VALU wave-instructions per launch of each (sum): %.3f M" % (tot / 1e6))
PY
cat gpurun_out/${tag}_pmc_all_kernels.log

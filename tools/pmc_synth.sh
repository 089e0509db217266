#!/bin/bash
# PMC passes for k_synth on the default bench workload (run on the GPU box from the repo root):
#   tools/pmc_synth.sh <tag>   ->  gpurun_out/<tag>_pmc_*.json  (copy the ones to be judged into profiles/)
# Counters are collected in their own rocprofv3 runs (no tracing options), a few per pass.
set -u
tag=${1:-rXX}
export TMPDIR=/tmp
out=gpurun_out/pmc_$tag
mkdir -p $out
cmd="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-fresh-plan --pipeline 1"
i=0
for ctrs in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" \
            "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_LDS_IDX_ACTIVE" "WRITE_SIZE" "FETCH_SIZE"; do
    i=$((i+1))
    # (WRITE_SIZE and FETCH_SIZE do not fit one pass on gfx950: rocprofv3 aborts and then hangs -> own passes,
    # and every pass under its own timeout)
    timeout 150 rocprofv3 --pmc $ctrs --output-format csv -d $out/p$i -- $cmd > $out/p$i.log 2>&1 || echo "pass $i ($ctrs) failed or timed out"
done
python3 - "$out" "$tag" <<'PY'
import csv, glob, json, sys, collections
out, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(out + "/p*/*/*counter_collection.csv"):
    per = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        if "k_synth" not in r["Kernel_Name"]:  # (k_synth or k_synth_g: whichever the default bench runs)
            continue
        per[r["Dispatch_Id"]][r["Counter_Name"]] = per[r["Dispatch_Id"]].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    for d in per.values():
        for k, v in d.items():
            acc[k].append(v)
res = {k: sum(v) / len(v) for k, v in acc.items()}
res["launches_averaged"] = {k: len(v) for k, v in acc.items()}
if "WRITE_SIZE" in res and "FETCH_SIZE" in res:
    # WRITE_SIZE is in KiB on gfx950 (tools/wrcal.hip: exact for a coalesced 1 GiB fill); FETCH_SIZE counts
    # 32-byte requests against a 64-byte unit there, i.e. the raw KiB figure is doubled (MI355X_MICROARCH guide,
    # HBM / rocprofv3 section) -- same correction as in profiles/archive/r01_pmc_write_fetch.md
    res["hbm_bytes_per_launch"] = int((res["WRITE_SIZE"] + 2.0 * res["FETCH_SIZE"]) * 1024)
# which build of the kernel this was measured on: bench.py and tests/test_bench_contract.py compare it with the tree's
import hashlib
res["kernel_source_sha256"] = hashlib.sha256(b"".join(open("galileo-sdr-sim_amd/csrc/" + f, "rb").read() for f in (
    "synth_group.hip", "synth_common.h", "synth_dev.h"))).hexdigest()
json.dump(res, open("gpurun_out/%s_pmc_k_synth_all.json" % tag, "w"), indent=1, sort_keys=True)
print(json.dumps(res, indent=1, sort_keys=True))
PY

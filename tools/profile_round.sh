#!/bin/bash
# Profiles to be judged, for one round tag (run on the GPU box from the repo root, e.g. via gpurun):
#   tools/profile_round.sh r02   ->  gpurun_out/<tag>_bench_kernel_stats.csv   rocprofv3 --kernel-trace --stats of the default bench command
#                                     gpurun_out/<tag>_bench_under_rocprof.log   its output (the JSON line is in it)
#                                     gpurun_out/<tag>_standalone_kernel_stats.csv  the same for `bench.py --pipeline 1` (k_synth alone)
#                                     gpurun_out/<tag>_pmc_k_synth_all.json      PMC passes (tools/pmc_synth.sh)
#                                     gpurun_out/<tag>_pmc_k_synth.json          HBM traffic summary read by bench.py
# Copy the ones to be judged into profiles/ afterwards.
set -u
tag=${1:-rXX}
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_$tag
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag/stats -- \
    python bench.py --no-cpu-baseline --no-extras --no-fresh-plan > gpurun_out/${tag}_bench_under_rocprof.log 2>&1
f=$(ls gpurun_out/prof_$tag/stats/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/${tag}_bench_kernel_stats.csv
# the kernel ALONE: one handle, so no two k_synth launches overlap and the average is a per-launch cost
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag/standalone -- \
    python bench.py --pipeline 1 --no-cpu-baseline --no-extras --no-fresh-plan > gpurun_out/${tag}_standalone_under_rocprof.log 2>&1
f=$(ls gpurun_out/prof_$tag/standalone/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/${tag}_standalone_kernel_stats.csv
tools/pmc_synth.sh $tag > gpurun_out/prof_$tag/pmc.log 2>&1
python3 - "$tag" <<'PY'
import json, sys
tag = sys.argv[1]
d = json.load(open("gpurun_out/%s_pmc_k_synth_all.json" % tag))
out = {"kernel": "k_synth_g<12,false>", "workload": "M-SYN12 1199x260000x12ch, chunk 1024 (16-sample groups)",
       "write_size_kib": d.get("WRITE_SIZE"), "fetch_size_kib_raw": d.get("FETCH_SIZE"),
       "fetch_correction": "x2 (gfx950, MI355X_MICROARCH.md HBM section)",
       "hbm_bytes_per_launch": d.get("hbm_bytes_per_launch"), "kernel_source_sha256": d.get("kernel_source_sha256"),
       "source": "tools/pmc_synth.sh %s: rocprofv3 --pmc WRITE_SIZE / --pmc FETCH_SIZE in separate passes over `bench.py --steps 3 "
                 "--warmup 1 --pipeline 1`, averaged over the launches; WRITE_SIZE unit calibrated with tools/wrcal.hip" % tag}
json.dump(out, open("gpurun_out/%s_pmc_k_synth.json" % tag, "w"), indent=1)
print(json.dumps(d, indent=1, sort_keys=True))
PY
head -4 gpurun_out/${tag}_bench_kernel_stats.csv
head -3 gpurun_out/${tag}_standalone_kernel_stats.csv
grep '"metric"' gpurun_out/${tag}_bench_under_rocprof.log | cut -c1-400

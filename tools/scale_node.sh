#!/bin/bash
# ONE command for the day an 8-GPU node exists.  No scaling claim is made anywhere in this repository: no
# multi-GPU node has reached this build in six rounds; this script produces the numbers and checks, it does not interpret them.
#
#   tools/scale_node.sh [tag]                on an N-GPU MI355X node (N = the visible devices; uses 1, 2, 4, 8 up to N)
#   GAL_BENCH_DEVICE=0 tools/scale_node.sh   REHEARSAL on a box with ONE GPU: every rank on device 0, gloo instead of RCCL for the
#                                            barriers and the report -- exercises every command below; every line it prints is
#                                            marked "rehearsal" and says nothing about scaling
# Output: gpurun_out/<tag>_scale_node.log (+ <tag>_scale_N<n>.json, one bench line per world size)
#   1. weak scaling, the headline: `bench.py --gpus N` for N = 1, 2, 4, 8 -- M-SYN12, one independent scenario per rank, no data-path
#      collective; per N: aggregate Msamples/s, ms per step, report backend (RCCL), rank count, rank_imbalance
#   1b. strong split: ONE M-SYN12 scenario cut into epoch ranges over N = 2, 4, 8 ranks (`--shard scenario`), per-rank walker / kernel time
#   2. config 5 literally: `bench.py --gpus 8 --workload locations` -- rank r = site r of shard.LOCATIONS, 300 s, host front-end rows
#   3. the product: `galileo-sdr-sim --sites` -- the eight sites over the node's devices, one ishort file per site, each file's md5
#      against THE REFERENCE PROGRAM's (tests/golden/ref_task_config5.json), per-site sink rate from the CLI's own lines
set -u
tag=${1:-r06}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
log=gpurun_out/${tag}_scale_node.log
rehearsal=""
if [ -n "${GAL_BENCH_DEVICE:-}" ]; then
    rehearsal="REHEARSAL (all ranks on GPU ${GAL_BENCH_DEVICE}, gloo): launch-path check, NOT a scaling measurement -- "
    export GAL_BENCH_BACKEND=${GAL_BENCH_BACKEND:-gloo}
    ndev=8
else
    ndev=$(python - <<'PY'
import torch
print(torch.cuda.device_count())
PY
)
fi
steps=${GAL_SCALE_STEPS:-40}
echo "### ${rehearsal}tools/scale_node.sh $tag: $ndev device(s), $(date -u +%FT%TZ)" | tee $log
port=29531
summ() { python - "$1" "$2" <<'PY'
import json, sys
path, what = sys.argv[1], sys.argv[2]
try:
    d = json.loads([ln for ln in open(path) if ln.startswith("{")][-1])
except Exception as e:
    print("%s: NO LINE (%s)" % (what, e)); sys.exit(0)
ranks = d.get("ranks") or []
print("%s: n_gpus %d  value %.1f Msamples/s  ms/step %.4f  scaling %s  report_backend %s  ranks reporting %d  rank_imbalance %s%s" % (
    what, d["n_gpus"], d["value"], d["ms_per_step"], d["scaling"], d.get("report_backend"), len(ranks) or 1, d.get("rank_imbalance"),
    "  [%s]" % d["rehearsal"] if d.get("rehearsal") else ""))
for r in ranks:
    print("    rank %d: epochs %s  walker chain %.4f ms  kernel %.4f ms  legs walked %d" % (r["rank"], r["epochs"], r["avg_walk_ms"], r["avg_kernel_ms"], r["legs_walked"]))
PY
}
# ---- 1. weak scaling of the engine, M-SYN12 per rank
for n in 1 2 4 8; do
    [ $n -gt $ndev ] && break
    out=gpurun_out/${tag}_scale_N${n}.json
    if [ $n -eq 1 ]; then
        timeout 900 python bench.py --gpus 1 --steps $steps --warmup 5 --no-extras --no-cpu-baseline --no-fresh-plan > $out 2>> $log.err
    else
        timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $port \
            bench.py --gpus $n --steps $steps --warmup 5 > $out 2>> $log.err
        port=$((port + 1))
    fi
    summ $out "weak M-SYN12 N=$n" | tee -a $log
done
# ---- 1b. ONE scenario cut into epoch ranges over the ranks (strong split, gal_synth_execute_range; no exchange): rank_imbalance is the
#         figure of interest -- shard.epoch_range's cost model was fitted on ranks run alone on one GPU (tools/strong_split_alone.sh)
for n in 2 4 8; do
    [ $n -gt $ndev ] && break
    out=gpurun_out/${tag}_scale_strong_N${n}.json
    timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $port \
        bench.py --gpus $n --steps $steps --warmup 5 --shard scenario > $out 2>> $log.err
    port=$((port + 1))
    summ $out "strong (one M-SYN12 scenario) N=$n" | tee -a $log
done
# ---- 2. config 5: eight sites, one per rank (as many ranks as there are devices; fewer ranks = the first sites)
n5=$ndev; [ $n5 -gt 8 ] && n5=8
if [ $n5 -ge 2 ]; then
    out=gpurun_out/${tag}_scale_locations_N${n5}.json
    timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n5 --master-addr 127.0.0.1 --master-port $port \
        bench.py --gpus $n5 --workload locations --steps 20 --warmup 3 > $out 2>> $log.err
    port=$((port + 1))
    summ $out "config 5 (locations) N=$n5" | tee -a $log
fi
# ---- 3. the product CLI over the node's devices: one process per site, files checked against the reference program's md5s
python - "$tag" "${GAL_BENCH_DEVICE:-}" <<'PY' 2>&1 | tee -a $log
import hashlib, json, os, re, shutil, subprocess, sys, tempfile, time
sys.path.insert(0, os.getcwd())
from __graft_entry__ import load_pkg
pkg = load_pkg()
tag, rehearsal = sys.argv[1], sys.argv[2] != ""
rec = json.load(open("tests/golden/ref_task_config5.json"))
cli = os.path.join("galileo-sdr-sim_amd", "galileo-sdr-sim")
nav = os.path.join("tests", "golden", "20feb2022.rnx")
base = "/dev/shm" if os.access("/dev/shm", os.W_OK) and shutil.disk_usage("/dev/shm").free > 30e9 else tempfile.gettempdir()
d = tempfile.mkdtemp(dir=base, prefix="galscale_")
try:
    lst = os.path.join(d, "sites.txt")
    open(lst, "w").write("".join("%.10g,%.10g,%.10g\n" % tuple(s["llh"]) for s in rec["sites"]))
    import torch
    ngpu = 1 if rehearsal else max(1, torch.cuda.device_count())
    per = (8 + ngpu - 1) // ngpu
    t0 = time.time()
    r = subprocess.run([cli, "-e", nav, "--sites", lst, "-t", rec["start"], "-d", str(rec["duration_s"]), "-o", os.path.join(d, "c5.ishort"),
                        "--gpus", str(ngpu), "--per-gpu", str(min(per, 2))], capture_output=True, text=True, timeout=3000)
    wall = time.time() - t0
    print("galileo-sdr-sim --sites: exit %d, %d device(s), wall %.1f s%s" % (r.returncode, ngpu, wall, "  [REHEARSAL on one GPU]" if rehearsal else ""))
    for ln in r.stderr.splitlines():
        if re.search(r"site|Sites|Msamples", ln):
            print("    " + ln.strip()[:200])
    bad = 0
    for k, s in enumerate(rec["sites"]):
        f = os.path.join(d, "c5.site%d.ishort" % k)
        h, n = hashlib.md5(), 0
        if os.path.exists(f):
            with open(f, "rb") as fh:
                for blk in iter(lambda: fh.read(1 << 24), b""):
                    h.update(blk); n += len(blk)
            os.remove(f)
        ok = (h.hexdigest(), n) == (s["md5"], s["bytes"])
        bad += not ok
        print("    site %d %s: %d bytes, md5 %s %s the reference program's" % (k, tuple(s["llh"]), n, h.hexdigest(), "==" if ok else "!="))
    print("galileo-sdr-sim --sites: %d of 8 files equal the reference program's" % (8 - bad))
finally:
    shutil.rmtree(d, ignore_errors=True)
PY
echo "### done: $log" | tee -a $log

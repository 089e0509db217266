#!/bin/bash
# Which HSA queue does each kernel of the pipelined bench land on?  rocprofv3 --kernel-trace (Queue_Id per dispatch), plain
# process vs. RCCL process group initialised first.   -> gpurun_out/<tag>_queue_map.log
tag=${1:-rXX}
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
args="--gpus 1 --steps 6 --warmup 2 --preroll-ms 0 --no-extras --no-cpu-baseline"
tr="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29521"
log=gpurun_out/${tag}_queue_map.log
: > $log
run() {
    name=$1; shift
    rm -rf /tmp/qm_$name
    "$@" > /tmp/qm_$name.out 2>&1
    f=$(find /tmp/qm_$name -name '*kernel_trace.csv' | head -1)
    echo "### $name" >> $log
    python3 - "$f" >> $log <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
qcol = [c for c in rows[0] if c.lower().replace("_", "") == "queueid"][0]
scol = [c for c in rows[0] if c.lower().replace("_", "") == "streamid"]
m = collections.defaultdict(collections.Counter)
for r in rows:
    n = r["Kernel_Name"].split("(")[0][:40]
    key = (r[qcol], r[scol[0]] if scol else "-")
    m[key][n] += 1
for k in sorted(m):
    print("queue %s stream %s: %s" % (k[0], k[1], dict(m[k])))
PY
}
run plain rocprofv3 --kernel-trace --output-format csv -d /tmp/qm_plain -- python bench.py $args
GAL_BENCH_FORCE_DIST=1 GAL_BENCH_PG_ORDER=early GAL_BENCH_RCCL=eager run rccl_early rocprofv3 --kernel-trace --output-format csv -d /tmp/qm_rccl_early -- $tr bench.py $args
GAL_BENCH_FORCE_DIST=1 run rccl_late rocprofv3 --kernel-trace --output-format csv -d /tmp/qm_rccl_late -- $tr bench.py $args
GPU_MAX_HW_QUEUES=8 run plain_q8 rocprofv3 --kernel-trace --output-format csv -d /tmp/qm_plain_q8 -- python bench.py $args
cat $log

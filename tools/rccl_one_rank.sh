#!/bin/bash
# The RCCL calls of bench.py's N > 1 path on a box with ONE GPU: world size 1 under torch.distributed.run, backend nccl
# (= RCCL): init_process_group(device_id), barrier, all_reduce MAX (f64) / SUM (i64), all_gather_object.  A launch-path check
# (tools/multiproc_one_gpu.sh covers 2 ranks with gloo); says nothing about scaling.   -> gpurun_out/<tag>_rccl_one_rank.log
tag=${1:-rXX}
export GAL_BENCH_FORCE_DIST=1 HSA_ENABLE_IPC_MODE_LEGACY=0
log=gpurun_out/${tag}_rccl_one_rank.log
mkdir -p gpurun_out
echo "### torchrun --nproc-per-node 1 bench.py --gpus 1 (GAL_BENCH_FORCE_DIST=1, backend nccl = RCCL)" > $log
for shard in scenarios scenario; do
    timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29519 \
        bench.py --gpus 1 --steps 20 --warmup 5 --shard $shard --no-extras --no-cpu-baseline 2>&1 | grep -v "^W\|^\*\*\*\|amdgpu.ids" >> $log
done
cut -c1-600 $log

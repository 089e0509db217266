#!/bin/bash
# Rehearsal of the multi-rank launch path on a 1-GPU box: 2 ranks, both on GPU 0, gloo backend.
# Writes gpurun_out/<tag>_multiproc_one_gpu.log -- a launch-path check, NOT a scaling measurement.
tag=${1:-rXX}
export GAL_BENCH_DEVICE=0 GAL_BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
log=gpurun_out/${tag}_multiproc_one_gpu.log
: > $log
for shard in scenarios scenario; do
    echo "### torchrun --nproc-per-node 2 bench.py --gpus 2 --shard $shard (both ranks on GPU 0, gloo)" >> $log
    timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
        bench.py --gpus 2 --steps 20 --warmup 3 --shard $shard 2>&1 | grep -v "^W\|^\*\*\*\|amdgpu.ids" >> $log
done
echo "### single process, same steps, for reference" >> $log
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2>&1 | grep -v amdgpu.ids >> $log
cat $log | cut -c1-700

#!/usr/bin/env python3
"""What does the end-of-region barrier cost with the RCCL process group (world size as launched)?
dist.barrier() vs an all_reduce on a resident one-element tensor + torch.cuda.synchronize()."""
import os
import time

import torch
import torch.distributed as dist

lr = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(lr)
dist.init_process_group(backend="nccl", device_id=torch.device("cuda", lr))
t = torch.zeros(1, device="cuda")


def timeit(fn, n=20):
    fn()
    torch.cuda.synchronize()
    v = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        v.append((time.perf_counter() - t0) * 1e6)
    v.sort()
    return v[len(v) // 2], v[0], v[-1]


def b1():
    dist.barrier()
    torch.cuda.synchronize()


def b2():
    dist.all_reduce(t)
    torch.cuda.synchronize()


def b3():
    dist.barrier(device_ids=[lr])
    torch.cuda.synchronize()


for name, fn in (("dist.barrier()", b1), ("all_reduce(resident tensor)", b2), ("dist.barrier(device_ids)", b3)):
    print("%-30s median %8.1f us  min %8.1f  max %8.1f" % ((name,) + timeit(fn)))
dist.destroy_process_group()

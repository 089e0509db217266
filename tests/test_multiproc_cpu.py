"""N>1 control path on CPU: two gloo ranks, each owning an independent scenario (weak-scaling split of
galileo-sdr-sim_amd/shard.py); the oracle stands in for the device so the reductions can be checked
against a single-process evaluation of both shards."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from __graft_entry__ import load_pkg
    from oracle_binding import oracle_run

    pkg = load_pkg()
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    r, lr, w = pkg.shard.dist_env()
    assert (r, lr, w) == (rank, rank, world)
    params = pkg.shard.rank_workload(rank, n_epochs=2, n_chan=3, n_slots=4, samples_per_epoch=2600)
    iq, _ = oracle_run(params, 2600, 2.6e6)
    chk = int(iq.astype(np.int64).sum()) & 0xFFFFFFFF
    dist.barrier()
    el, total, chk_all = pkg.shard.reduce_report(dist, "cpu", 0.5 + rank, iq.size // 2, chk)
    q.put((rank, el, total, chk_all, chk))
    dist.destroy_process_group()


def test_two_rank_weak_scaling_split(pkg):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # every rank sees the same reduced report: MAX time, SUM samples, SUM checksums
    assert res[0][1] == res[1][1] == 1.5
    assert res[0][2] == res[1][2] == 2 * 2 * 2600
    assert res[0][3] == res[1][3] == (res[0][4] + res[1][4]) & 0xFFFFFFFF
    # shards are different scenarios of identical size
    a = pkg.shard.rank_workload(0, 2, 3, 4, 2600)
    b = pkg.shard.rank_workload(1, 2, 3, 4, 2600)
    assert a.shape == b.shape and a.tobytes() != b.tobytes()

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
    # the same through a second group of the same ranks: bench.py's barriers (and, should RCCL fail, its report) go through
    # a gloo group beside the default one
    ctl = dist.new_group(backend="gloo")
    dist.barrier(group=ctl)
    assert pkg.shard.reduce_report(dist, "cpu", 0.5 + rank, iq.size // 2, chk, group=ctl) == (el, total, chk_all)
    got = [None] * world
    dist.all_gather_object(got, {"rank": rank}, group=ctl)
    assert [g["rank"] for g in got] == list(range(world))
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


def _worker_strong(rank, world, port, q):
    """One scenario cut into epoch ranges: each rank evaluates its range with the carrier state the chain has at
    its first epoch (on the GPU the engine's own walker provides it; here the oracle is run over the preceding
    epochs and its end state handed over, which is the same statement)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from __graft_entry__ import load_pkg
    from oracle_binding import oracle_run

    pkg = load_pkg()
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    params = pkg.shard.rank_workload(0, n_epochs=7, n_chan=3, n_slots=4, samples_per_epoch=2600)  # the SAME scenario
    e0, ne = pkg.shard.epoch_range(rank, world, 7)
    state = None
    if e0 > 0:
        _, state = oracle_run(params[:e0], 2600, 2.6e6)
    mine = params[e0:e0 + ne].copy()
    if e0 > 0:
        mine["flags"][0, :] = 0  # continues from the carried state
    iq, _ = oracle_run(mine, 2600, 2.6e6, state)
    chk = int(iq.astype(np.int64).sum()) & 0xFFFFFFFF
    dist.barrier()
    el, total, chk_all = pkg.shard.reduce_report(dist, "cpu", 1.0, iq.size // 2, chk)
    q.put((rank, e0, ne, total, chk_all))
    dist.destroy_process_group()


def test_two_rank_strong_scaling_split(pkg):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_binding import oracle_run

    for world, n in [(1, 7), (2, 7), (3, 7), (4, 10), (8, 1199), (8, 8), (8, 5), (4, 2999)]:
        for w, pp in ((0.0, 0.0), (pkg.shard.WALK_COST, 0.0), (0.5, 0.0), (pkg.shard.WALK_COST, pkg.shard.PREFIX_PASS_COST)):
            r = [pkg.shard.epoch_range(k, world, n, walk_cost=w, prefix_pass_cost=pp) for k in range(world)]
            assert r[0][0] == 0 and sum(c for _, c in r) == n and all(r[k][0] + r[k][1] == r[k + 1][0] for k in range(world - 1))
            assert all(c >= (1 if n >= world else 0) for _, c in r)
            if w == 0.0:
                assert max(c for _, c in r) - min(c for _, c in r) <= 1
            elif n >= 1000:
                # the model's cost is level: count + walk_cost x prefix + (prefix ? prefix_pass_cost : 0); later ranks get shorter ranges
                cost = [c + (w * a + pp if k else 0.0) for k, (a, c) in enumerate(r)]
                assert max(cost) - min(cost) <= 2.0 * (1.0 + w) and all(r[k][1] >= r[k + 1][1] for k in range(world - 1)), (world, n, w, pp, r)
    # the default split of the 120 s scenario over 8 ranks: rank 0, which has no prefix and needs no second walker pass, takes a third
    assert [pkg.shard.epoch_range(k, 8, 1199) for k in (0, 1, 7)] == [(0, 443), (443, 171), (1139, 60)]
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_strong, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # (7 epochs: a prefix costs its rank a second walker pass worth 200 epochs of synthesis, so rank 0 takes all it can)
    assert (res[0][1], res[0][2], res[1][1], res[1][2]) == (0, 6, 6, 1)
    full, _ = oracle_run(pkg.shard.rank_workload(0, 7, 3, 4, 2600), 2600, 2.6e6)
    assert res[0][3] == res[1][3] == 7 * 2600
    assert res[0][4] == res[1][4] == int(full.astype(np.int64).sum()) & 0xFFFFFFFF


def test_config5_locations_split(pkg):
    """BASELINE config 5 literally: rank r owns static site r (shard.LOCATIONS) for the same duration; the sites see
    different satellites, so the scenarios differ while their size in samples is the same."""
    nav = os.path.join(ROOT, "tests", "golden", "20feb2022.rnx")
    rows = [pkg.shard.rank_location_scenario(pkg.Scenario, nav, r, duration_s=5.0)[0] for r in range(8)]
    assert all(x.shape == (49, 16) for x in rows)
    counts = [int((x["prn"][0] > 0).sum()) for x in rows]
    assert counts == [9, 10, 9, 9, 5, 6, 6, 9]
    assert len({x.tobytes() for x in rows}) == 8
    # rank 8 wraps around to site 0
    again, llh = pkg.shard.rank_location_scenario(pkg.Scenario, nav, 8, duration_s=5.0)
    assert llh == pkg.shard.LOCATIONS[0] and again.tobytes() == rows[0].tobytes()

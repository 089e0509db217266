"""Work split across GPUs of one node: the path shards embarrassingly -- every rank
synthesises its own independent scenario (BASELINE config 5: N locations x duration), there is no
data-path collective.  torch.distributed (RCCL on the GPU box, gloo in CPU tests) carries only the
barrier and the tiny report reductions."""
import os

from . import workloads


def dist_env():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def rank_workload(rank, n_epochs, n_chan=12, n_slots=16, samples_per_epoch=260000, sample_rate=2.6e6, dyn_track=False):
    """The scenario rank `rank` owns: same size on every rank (weak scaling), different seed."""
    return workloads.make_synthetic(n_epochs=n_epochs, n_chan=n_chan, n_slots=n_slots,
                                    samples_per_epoch=samples_per_epoch, sample_rate=sample_rate,
                                    seed=workloads.SEED + rank, dyn_track=dyn_track)


# BASELINE config 5, literally: "8 independent static locations x 300 s sharded across 8 x MI355X".  Eight receiver sites
# (lat [deg], lon [deg], height [m]); rank r simulates LOCATIONS[r % 8] from the navigation file through the real host
# front-end (RINEX -> orbits -> ranges -> I/NAV), so the ranks see different satellites and different SV counts.
LOCATIONS = (
    (-6.0, 51.0, 100.0),     # the survey's anchor site (Indian Ocean)
    (45.0, 10.0, 100.0),     # northern Italy: 10 SVs with 20feb2022.rnx
    (0.0, 0.0, 100.0),       # Gulf of Guinea
    (60.0, 25.0, 100.0),     # Helsinki
    (42.3601, -71.0589, 2.0),  # the reference's default (src/main.cpp:179-196)
    (35.274, 137.014, 100.0),  # the reference's usage example
    (-33.9, 18.4, 50.0),     # Cape Town
    (52.0, 4.4, 10.0),       # Delft
)


def rank_location_scenario(scenario_cls, nav_file, rank, duration_s=300.0, start="2022/02/20,12:00:00", n_slots=16):
    """[n_epochs, n_slots] rows of the static scenario rank `rank` owns in the config-5 split (host front-end)."""
    llh = LOCATIONS[rank % len(LOCATIONS)]
    return scenario_cls(nav_file, llh=llh, start=start, duration_s=duration_s, iono_enable=True, n_slots=n_slots).all(), llh


WALK_COST = 0.16  # walking an epoch's NCO chains silently, relative to synthesising it (VALU work: walkers 37 M against
                 # k_synth_g's 417 M wave-instructions per 1199 epochs = 0.09; measured, ranks run alone on one GPU: +0.000145 ms of
                 # walker chain per prefix epoch against 0.00088 ms of synthesis per epoch, profiles/r05n_strong_split_alone.log)
PREFIX_PASS_COST = 200.0  # ... and what ANY prefix costs on top, in epochs of synthesis: legs in front of the executed range are never
                          # accepted by translation (their checkpoints do not exist, nothing could check them), so a rank with a prefix
                          # runs a second walker pass: 0.40 ms of chain at prefix 0 against 0.22 (same log) = 0.18 ms = 200 epochs of k_synth_g


def epoch_range(rank, world, n_epochs, walk_cost=WALK_COST, prefix_pass_cost=PREFIX_PASS_COST):
    """Contiguous epoch range [first, first + count) of ONE scenario for rank `rank` (strong scaling, SURVEY.md
    §8e-ii): every rank plans the whole scenario and synthesises only its own range (gal_synth_execute_range).  The
    carrier chain never restarts, so a rank WALKS the epochs [0, first + count) -- its prefix silently -- and the ranges are
    cut so that walk + synth is the same on every rank.  Cost model, in epochs of synthesis: rank 0 pays its count; a rank with a
    prefix b pays count + walk_cost * b + prefix_pass_cost (the second walker pass its prefix legs need).  Rank 0 therefore gets
    the longest range, and the later ranks shorter and shorter ones.  (walk_cost = 0 and prefix_pass_cost = 0: equal ranges.)
    Round 4's model had the proportional term only: ranks measured ALONE on one GPU were 16 % / 15 % / 9 % out of balance at
    world 2 / 4 / 8 (tools/strong_split_alone.sh; with this model 1.09 / 1.06 / 1.07 at its first constants, profiles/r05n_*)."""
    if world == 1:
        return 0, n_epochs

    def counts(T):
        out, b = [], 0.0
        for r in range(world):
            n = T if r == 0 else T - prefix_pass_cost - walk_cost * b
            n = max(n, 0.0)
            out.append(n)
            b += n
        return out

    if walk_cost <= 0.0 and prefix_pass_cost <= 0.0:
        b = [(k * n_epochs) // world for k in range(world + 1)]
    else:
        lo, hi = 0.0, float(n_epochs) + prefix_pass_cost + 1.0
        for _ in range(80):  # the level T at which the ranges add up to the scenario
            mid = 0.5 * (lo + hi)
            if sum(counts(mid)) < n_epochs:
                lo = mid
            else:
                hi = mid
        c = counts(hi)
        b = [0]
        acc = 0.0
        for r in range(world):
            acc += c[r]
            b.append(int(round(acc)))
        m = 1 if n_epochs >= world else 0  # every rank at least one epoch where there are enough of them
        b[0] = 0
        for k in range(1, world):
            b[k] = max(b[k], b[k - 1] + m)
        b[world] = n_epochs
        for k in range(world - 1, 0, -1):
            b[k] = min(b[k], b[k + 1] - m)
    return b[rank], b[rank + 1] - b[rank]


def reduce_report(dist, device, elapsed_s, n_samples, checksum, group=None):
    """MAX of the elapsed time, SUM of samples, XOR-free SUM of 32-bit checksums over ranks (`group`: None = the default
    process group; tensors live on `device`, which must suit the group's backend)."""
    import torch

    if dist is None:
        return elapsed_s, n_samples, checksum & 0xFFFFFFFF
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    c = torch.tensor([n_samples, checksum & 0xFFFFFFFF], dtype=torch.int64, device=device)
    dist.all_reduce(c, op=dist.ReduceOp.SUM, group=group)
    return float(t.item()), int(c[0].item()), int(c[1].item()) & 0xFFFFFFFF

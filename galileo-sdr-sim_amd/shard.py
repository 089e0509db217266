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


WALK_COST = 0.1  # walking an epoch's NCO chains silently, relative to synthesising it (VALU work: walkers 37 M against
                 # k_synth_g's 429 M wave-instructions per 1199 epochs, and legs in front of a range may be walked twice)


def epoch_range(rank, world, n_epochs, walk_cost=WALK_COST):
    """Contiguous epoch range [first, first + count) of ONE scenario for rank `rank` (strong scaling, SURVEY.md
    §8e-ii): every rank plans the whole scenario and synthesises only its own range (gal_synth_execute_range).  The
    carrier chain never restarts, so a rank WALKS the epochs [0, first + count) -- its prefix silently -- and the ranges are
    cut so that walk(prefix + range) + synth(range) is the same on every rank: with w = walk_cost the boundaries satisfy
    w b[r+1] + (b[r+1] - b[r]) = const, i.e. later ranks get shorter ranges (w = 0: equal ranges)."""
    def bounds():
        if walk_cost <= 0.0 or world == 1:
            return [(k * n_epochs) // world for k in range(world + 1)]
        q = 1.0 / (1.0 + walk_cost)
        total = 1.0 - q ** world
        b = [int(round(n_epochs * (1.0 - q ** k) / total)) for k in range(world + 1)]
        m = 1 if n_epochs >= world else 0  # every rank at least one epoch where there are enough of them
        b[0] = 0
        for k in range(1, world):
            b[k] = max(b[k], b[k - 1] + m)
        b[world] = n_epochs
        for k in range(world - 1, 0, -1):
            b[k] = min(b[k], b[k + 1] - m)
        return b

    b = bounds()
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

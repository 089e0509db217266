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


def epoch_range(rank, world, n_epochs):
    """Contiguous epoch range [first, first + count) of ONE scenario for rank `rank` (strong scaling, SURVEY.md
    §8e-ii): every rank plans the whole scenario -- the NCO walk over all epochs is what gives it the exact
    carrier state at its first epoch -- and synthesises only its own range (gal_synth_execute_range)."""
    base, extra = divmod(n_epochs, world)
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


def reduce_report(dist, device, elapsed_s, n_samples, checksum):
    """MAX of the elapsed time, SUM of samples, XOR-free SUM of 32-bit checksums over ranks."""
    import torch

    if dist is None:
        return elapsed_s, n_samples, checksum & 0xFFFFFFFF
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    c = torch.tensor([n_samples, checksum & 0xFFFFFFFF], dtype=torch.int64, device=device)
    dist.all_reduce(c, op=dist.ReduceOp.SUM)
    return float(t.item()), int(c[0].item()), int(c[1].item()) & 0xFFFFFFFF

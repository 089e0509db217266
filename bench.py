#!/usr/bin/env python3
"""bench.py -- headline benchmark of the Galileo E1B/C IQ synthesis hot path on MI355X.

One "step" = one pass of the hot path (NCO walk + per-sample synthesis, reference
src/galileo-sdr.cpp:481-539) over one batch of synthetic input already resident in HBM:
workload M-SYN12 of SURVEY.md §8(d) = BASELINE.json configs[1] geometry (static, 12 SVs, 120 s at
2.6 MS/s -> 1199 epochs x 260000 samples = 311.74 M complex samples = 1.247 GB of int16 IQ).

    python bench.py --gpus N --steps K --warmup W
Default: two engine handles in flight (software pipeline, see --pipeline), 100 timed steps.
N > 1 is launched by the driver as `python -m torch.distributed.run --nproc-per-node N ... bench.py`;
one process per GPU, each rank synthesises its OWN independent scenario of the same size (weak scaling,
no data-path collective: scenarios shard embarrassingly); torch.distributed (RCCL)
is used only for the barrier and the max-over-ranks time.

Rank 0 prints ONE JSON line (see README / DESIGN.md for the field meanings).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
METRIC = "IQ Msamples/s (×real-time @2.6MS/s), 12-SV static E1B/C, 1/2/4/8 GPU"


def cpu_baseline(pkg, params, n_samp, rate, target_seconds=12.0):
    """Time the CPU restatement of the reference loop (oracle/galsyn_oracle.c, 1 thread, -O2
    -ffp-contract=off) on a bounded prefix of the same workload.  Checker code used ONLY as the
    reported baseline, never in the product path."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_binding import oracle_run

    probe = 4
    t0 = time.perf_counter()
    oracle_run(params[:probe], n_samp, rate)
    dt = time.perf_counter() - t0
    n_ep = int(max(probe, min(params.shape[0], target_seconds / max(dt / probe, 1e-6))))
    t0 = time.perf_counter()
    oracle_run(params[:n_ep], n_samp, rate)
    dt = time.perf_counter() - t0
    plain = n_ep * n_samp / dt / 1e6
    # with the reference's per-sample clock read (src/galileo-sdr.cpp:485), on a shorter prefix
    n_ck = max(probe, n_ep // 4)
    t0 = time.perf_counter()
    oracle_run(params[:n_ck], n_samp, rate, clock_read=True)
    dtc = time.perf_counter() - t0
    with_clock = n_ck * n_samp / dtc / 1e6
    # many host cores, one independent scenario slice per core (the reference itself is single-threaded);
    # bounded: <= 64 threads x 3 x 48 epochs (each call allocates its own 50 MB of IQ)
    from concurrent.futures import ThreadPoolExecutor

    cores = min(os.cpu_count() or 1, 64)
    n_par, reps = min(48, params.shape[0]), 3

    def work(_):
        for _r in range(reps):
            oracle_run(params[:n_par], n_samp, rate)

    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:  # ctypes releases the GIL inside gal_oracle_run
        list(ex.map(work, range(cores)))
    dtp = time.perf_counter() - t0
    all_cores = cores * reps * n_par * n_samp / dtp / 1e6
    n_act = int((params["prn"][0] > 0).sum())
    return {
        "value": round(plain, 3),
        "unit": "Msamples/s",
        "cores": 1,
        "kind": "port",
        "sample": "first %d of %d epochs of the same workload (%d SVs, %d samples/epoch), %.1f s of CPU; "
        "with the reference's per-sample get_nanos(): %.3f Msamples/s; %d independent scenario slices on %d cores: "
        "%.1f Msamples/s aggregate" % (n_ep, params.shape[0], n_act, n_samp, dt, with_clock, cores, cores, all_cores),
    }


def measured_traffic():
    """HBM bytes per k_synth launch from the newest committed PMC summary (rocprofv3 --pmc WRITE_SIZE /
    FETCH_SIZE passes, profiles/*_pmc_k_synth.json); None if there is none.  bench.py cannot run
    rocprofv3 on itself, so this is the last profiled value for the same workload, not a live one."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_k_synth.json")))
    if not files:
        return None, None
    try:
        d = json.load(open(files[-1]))
        return int(d["hbm_bytes_per_launch"]), os.path.basename(files[-1])
    except Exception:
        return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100, help="timed steps (1.6 ms each; the pipeline is empty at both "
                    "ends of the timed region, so few steps mostly measure its fill and drain)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--epochs", type=int, default=1199, help="epochs per step (default: the 120 s scenario)")
    ap.add_argument("--channels", type=int, default=12)
    ap.add_argument("--chunk", type=int, default=0, help="samples per lane (0 = auto)")
    ap.add_argument("--workload", default="syn12", choices=["syn12", "syn24", "dyn"],
                    help="syn12 = headline M-SYN12 (BASELINE configs[1] size); syn24 = config 4 geometry (24 SVs, "
                    "25 MS/s); dyn = config 3 (12 SVs, Doppler from a 10 Hz circular track)")
    ap.add_argument("--pipeline", type=int, default=2, choices=[1, 2, 3, 4],
                    help="engine handles in flight: 2 = software pipeline, the NCO walk of step k+1 (latency "
                    "bound, on the handle's high-priority stream) runs beside the synthesis kernel of step k (issue "
                    "bound); every step still does the complete pass into its own buffers (measured: 1 / 2 / 3 "
                    "handles = 2.19 / 1.67 / 1.85 ms per step)")
    ap.add_argument("--shard", default="scenarios", choices=["scenarios", "scenario"],
                    help="N > 1: 'scenarios' = one independent scenario per rank (weak scaling, the default and the "
                    "headline); 'scenario' = ONE scenario cut into contiguous epoch ranges, every rank walks the whole "
                    "NCO chain and synthesises its own range (strong scaling, no exchange)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the synthesis engine has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod

        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    from __graft_entry__ import load_pkg

    pkg = load_pkg()
    n_samp, rate, n_slots = 260000, 2.6e6, 16
    if args.workload == "syn24":
        n_samp, rate, n_slots = 2500000, 25e6, 24
        args.channels = 24
    # each rank: an independent scenario of identical size (different seed)
    strong = args.shard == "scenario"
    params = pkg.shard.rank_workload(0 if strong else rank, args.epochs, n_chan=args.channels, n_slots=n_slots,
                                     samples_per_epoch=n_samp, sample_rate=rate, dyn_track=(args.workload == "dyn"))
    e_first, e_count = pkg.shard.epoch_range(rank, world, args.epochs) if strong else (0, args.epochs)
    depth = args.pipeline
    engines, outs, streams = [], [], []
    for k in range(depth):
        eng = pkg.SynthEngine(sample_rate=rate, samples_per_epoch=n_samp, n_slots=n_slots, device=local_rank,
                              chunk_samples=args.chunk)
        st = torch.cuda.Stream()
        eng.set_stream(st.cuda_stream)
        eng.plan(params)  # inputs resident in HBM before the timed region
        engines.append(eng)
        streams.append(st)
        outs.append(torch.empty(e_count * n_samp * 2, dtype=torch.int16, device="cuda"))
    out = outs[0]

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def run(n_steps):
        """n_steps complete passes, `depth` of them in flight; returns the per-step stats."""
        all_stats = []
        inflight = [False] * depth
        for k in range(n_steps):
            j = k % depth
            if inflight[j]:
                all_stats.append(engines[j].finish()[1])
            engines[j].execute(outs[j].data_ptr(), e_first, e_count)
            inflight[j] = True
        for k in range(n_steps, n_steps + depth):
            j = k % depth
            if inflight[j]:
                all_stats.append(engines[j].finish()[1])
                inflight[j] = False
        return all_stats

    run(args.warmup)
    barrier()
    t0 = time.perf_counter()
    step_stats = run(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    assert len(step_stats) == args.steps
    stats = step_stats[-1]
    ms_synth = sum(s["ms_synth"] for s in step_stats)
    ms_walk = sum(s["ms_walk"] for s in step_stats)
    assert all(s["chain_mismatch"] == 0 for s in step_stats)
    # outside the timed region: the same kernel without a co-running walker (one handle), for the roofline note
    solo_ms = None
    if depth > 1:
        inflight_stats = []
        for _ in range(3):
            engines[0].execute(outs[0].data_ptr(), e_first, e_count)
            inflight_stats.append(engines[0].finish()[1])
        solo_ms = sum(x["ms_synth"] for x in inflight_stats) / len(inflight_stats)
    samples_per_step = e_count * n_samp  # this rank's share
    # integrity of what was timed: a checksum of the last output (outside the timed region)
    chk = 0
    v32 = out.view(torch.int32)
    for a in range(0, v32.numel(), 1 << 28):  # in pieces: the int64 widening of a 60 GB output is 120 GB
        chk = (chk + int(v32[a:a + (1 << 28)].to(torch.int64).sum().item())) & 0xFFFFFFFF
    elapsed, total_samples, chk = pkg.shard.reduce_report(dist, "cuda", elapsed, samples_per_step * args.steps, chk)
    value = total_samples / elapsed / 1e6

    if rank == 0:
        avg_synth_ms = ms_synth / args.steps
        traffic, traffic_src = measured_traffic() if (args.epochs == 1199 and args.workload == "syn12" and args.channels == 12
                                                          and e_count == args.epochs) else (None, None)
        achieved = 4.0 * samples_per_step / (avg_synth_ms * 1e-3) / 1e9 if avg_synth_ms > 0 else 0.0
        line = {
            "metric": METRIC,
            "value": round(value, 3),
            "unit": "Msamples/s",
            "x_realtime": round(value / 2.6, 2),
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "f64 phase NCO -> int32 accumulate -> int16 IQ",
            "data": "synthetic",
            "config": {
                "workload": {"syn12": "M-SYN12: static-geometry 12-SV E1B/C", "syn24": "M-SYN24: 24-SV E1B/C",
                             "dyn": "M-DYN: 12-SV E1B/C, 10 Hz circular user motion"}[args.workload]
                + (", ONE scenario of %d epochs x %d samples @%.1f MS/s cut into epoch ranges over the ranks" if strong
                   else ", %d epochs x %d samples @%.1f MS/s per GPU (one independent scenario per rank)") % (
                    args.epochs, n_samp, rate / 1e6),
                "channels": args.channels,
                "chunk_samples": stats["chunk_samples"],
                "walk_passes": stats["walk_passes"],
                "chain_mismatch": stats["chain_mismatch"],
                "pipeline_depth": depth,
                "output_checksum": "%08x" % chk,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "k_synth<%d,false>%s" % (min(args.channels, 12), " (+ accumulate launch)" if args.channels > 12 else ""),
                "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5),
                "traffic": traffic,
                "traffic_source": traffic_src,
                "avg_kernel_ms": round(avg_synth_ms, 4),
                "note": "kernel duration inside the pipelined timed region (a walker of the next step co-runs)",
                "standalone_kernel_ms": round(solo_ms, 4) if solo_ms else None,
                "standalone_achieved": round(4.0 * samples_per_step / (solo_ms * 1e-3) / 1e9, 2) if solo_ms else None,
                "avg_walk_ms": round(ms_walk / args.steps, 4),
                "algorithmic_bytes_per_launch": 4 * samples_per_step,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(pkg, params, n_samp, rate)
        line["x_realtime"] = round(value * 1e6 / rate, 2)
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

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


def cpu_baseline(pkg, params, n_samp, rate, target_seconds=12.0, gpu_out=None, extra_checks=()):
    """Time the CPU restatement of the reference loop (oracle/galsyn_oracle.c, 1 thread, -O2
    -ffp-contract=off) on a bounded prefix of the same workload.  Checker code used ONLY as the
    reported baseline, never in the product path.  (The ONLY function of this file that touches anything under oracle/:
    tests/test_abi.py::test_product_never_touches_the_oracle checks that.)
    extra_checks: (label, params, device tensor) of other legs' outputs -- configs.fresh_plan's last two scenarios -- compared with the
    oracle here, every int16 of every epoch, on threads of their own; the verdicts come back under "extra_checks".
    gpu_out: the int16 output of the timed steps (device tensor): the oracle's samples over the epochs it ran -- all of them
    on the GPU box's host -- are compared with it int16 by int16; the verdict goes into the line."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_binding import oracle_run

    extra_verdicts = []
    if extra_checks:
        from concurrent.futures import ThreadPoolExecutor as _Pool

        def _check(item):
            label, q, dev = item
            ref, _ = oracle_run(q, n_samp, rate)
            bad_, piece_ = 0, 64 * n_samp * 2
            for a_ in range(0, ref.size, piece_):
                bad_ += int(np.count_nonzero(dev[a_:a_ + piece_].cpu().numpy() != ref[a_:a_ + piece_]))
            return {**label, "epochs_compared": int(q.shape[0]), "int16_different": bad_}

        with _Pool(len(extra_checks)) as ex_:  # ctypes releases the GIL inside the oracle
            extra_verdicts = list(ex_.map(_check, extra_checks))
    probe = 4
    t0 = time.perf_counter()
    oracle_run(params[:probe], n_samp, rate)
    dt = time.perf_counter() - t0
    n_ep = int(max(probe, min(params.shape[0], target_seconds / max(dt / probe, 1e-6))))
    t0 = time.perf_counter()
    ref_iq, _ = oracle_run(params[:n_ep], n_samp, rate)
    dt = time.perf_counter() - t0
    plain = n_ep * n_samp / dt / 1e6
    verdict = None
    if gpu_out is not None:
        # the timed output against the checker, every int16 of the epochs the checker produced (in pieces of 64 epochs)
        bad, piece = 0, 64 * n_samp * 2
        chk = 0
        for a in range(0, n_ep * n_samp * 2, piece):
            b = min(a + piece, n_ep * n_samp * 2)
            bad += int(np.count_nonzero(gpu_out[a:b].cpu().numpy() != ref_iq[a:b]))
            chk = (chk + int(ref_iq[a:b].view(np.int32).astype(np.int64).sum())) & 0xFFFFFFFF
        verdict = {"equal": bad == 0, "epochs_compared": int(n_ep), "of_epochs": int(params.shape[0]), "int16_different": bad,
                   "oracle_checksum": "%08x" % chk}
    del ref_iq
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
    # the reference's OWN loop: oracle/_ref/libref_loop.so = src/galileo-sdr.cpp:481-539 compiled from the reference's text with
    # the reference's flags (-g -DDEBUG, no -O; oracle/ref_loop_harness.cpp), where it was built (this needs /root/reference at
    # build time; the .so travels to the GPU box).  Its per-sample get_nanos() is the harness's no-op, so this is an upper bound.
    ref_own = None
    try:
        from ref_loop_binding import ref_loop_available, ref_loop_run
        if ref_loop_available() and rate == 2.6e6 and params.shape[1] <= 16:
            n_rl = max(2, min(params.shape[0], n_ep // 8))
            t0 = time.perf_counter()
            ref_loop_run(params[:n_rl], n_samp)
            dtr = time.perf_counter() - t0
            ref_own = {"value": round(n_rl * n_samp / dtr / 1e6, 3), "unit": "Msamples/s", "cores": 1, "kind": "reference",
                       "sample": "first %d epochs, %.1f s of CPU: the reference's loop text (src/galileo-sdr.cpp:481-539) compiled with "
                                 "its own flags (-g -DDEBUG, no -O), clock read stubbed" % (n_rl, dtr)}
    except Exception as e:  # a baseline, not the product: report and go on
        ref_own = {"error": str(e)[:200]}
    return {
        **({"extra_checks": extra_verdicts} if extra_verdicts else {}),
        **({"reference_loop": ref_own} if ref_own else {}),
        **({"output_vs_oracle": verdict} if verdict else {}),
        "value": round(plain, 3),
        "unit": "Msamples/s",
        "cores": 1,
        "kind": "port",
        "sample": "first %d of %d epochs of the same workload (%d SVs, %d samples/epoch), %.1f s of CPU; "
        "with the reference's per-sample get_nanos(): %.3f Msamples/s; %d independent scenario slices on %d cores: "
        "%.1f Msamples/s aggregate" % (n_ep, params.shape[0], n_act, n_samp, dt, with_clock, cores, cores, all_cores),
    }


def leg_kernel_plus_d2h(torch, engines, outs, streams, e_first, e_count, n_samp, steps=8):
    """SURVEY.md 8(d), second timing: the same pass PLUS the device->host copy of the IQ into pinned memory (what a
    caller that hands over HOST buffers pays; PCIe-inclusive, never the headline).  Double-buffered as in the CLI:
    the copy of step k runs beside the synthesis of step k+1."""
    depth = len(engines)
    host = [torch.empty(outs[0].numel(), dtype=torch.int16, pin_memory=True) for _ in range(depth)]
    copy_stream = torch.cuda.Stream()
    copied = [None] * depth

    def run(n):
        inflight = [False] * depth
        for k in range(n + depth):
            j = k % depth
            if inflight[j]:
                engines[j].finish()  # IQ final only after finish()
                copy_stream.wait_stream(streams[j])
                with torch.cuda.stream(copy_stream):
                    host[j].copy_(outs[j], non_blocking=True)
                    copied[j] = torch.cuda.Event()
                    copied[j].record(copy_stream)
                inflight[j] = False
            if k < n:
                if copied[j] is not None:
                    copied[j].synchronize()  # the device buffer is free again
                engines[j].execute(outs[j].data_ptr(), e_first, e_count)
                inflight[j] = True
        copy_stream.synchronize()

    run(2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"value": round(e_count * n_samp * steps / dt / 1e6, 1), "unit": "Msamples/s", "ms_per_step": round(dt / steps * 1e3, 3),
            "d2h_GBps": round(4.0 * e_count * n_samp * steps / dt / 1e9, 1), "steps": steps,
            "what": "execute + finish + copy into pinned host memory, double-buffered (PCIe-inclusive)"}


def leg_file_sink(pkg, seconds=120):
    """SURVEY.md 8(d), third timing: the drop-in itself -- the CLI with the reference's option surface on the RINEX
    scenario of BASELINE configs[0/1] (9 SVs with this navigation file), RINEX parsing, geodesy and I/NAV included,
    writing the ishort stream to /dev/null and to a tmpfs file.  The file is then hashed and compared with the same scenario
    through the library (front-end rows -> gal_synth_run_host, outside every timed region): what the timed command wrote is
    the stream the C-ABI produces (src/galileo-sdr.cpp:438,542,658: (10 d - 1) epochs of 260000 int16 pairs)."""
    import hashlib
    import re
    import subprocess
    import tempfile

    exe = os.path.join(ROOT, "galileo-sdr-sim_amd", "galileo-sdr-sim")
    nav = os.path.join(ROOT, "tests", "golden", "20feb2022.rnx")
    if not (os.path.exists(exe) and os.path.exists(nav)):
        return None
    out = {}
    tmpdir = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
    for name, sink in (("dev_null", "/dev/null"), ("tmpfs_file", os.path.join(tmpdir, "galbench_%d.ishort" % os.getpid()))):
        try:
            t0 = time.perf_counter()
            res = subprocess.run([exe, "-e", nav, "-l", "-6,51,100", "-t", "2022/02/20,12:00:00", "-d", str(seconds), "-U", "1",
                                  "-b", "1", "-P", "0", "-o", sink], capture_output=True, text=True, timeout=300)
            wall = time.perf_counter() - t0
            m = re.search(r"Process time = ([0-9.]+)", res.stderr)
            n_samples = (int(seconds * 10 + 0.5) - 1) * 260000
            if res.returncode != 0 or not m:
                out[name] = {"error": (res.stderr or res.stdout)[-200:]}
                continue
            # %.2f of the CLI's own line is too coarse for a 0.1 s run: use its Msamples/s figure
            m2 = re.search(r"\(([0-9.]+) Msamples/s", res.stderr)
            out[name] = {"value": float(m2.group(1)) if m2 else round(n_samples / float(m.group(1)) / 1e6, 1), "unit": "Msamples/s",
                         "process_time_s": float(m.group(1)), "wall_s_incl_startup": round(wall, 3),
                         "bytes": n_samples * 4}
            if sink != "/dev/null":
                h = hashlib.md5()
                with open(sink, "rb") as fh:
                    for blk in iter(lambda: fh.read(1 << 24), b""):
                        h.update(blk)
                rows = pkg.Scenario(nav, llh=(-6.0, 51.0, 100.0), start="2022/02/20,12:00:00", duration_s=float(seconds),
                                    iono_enable=True).all()
                with pkg.SynthEngine(device=-1) as eng:
                    iq, _, _ = eng.run_host(rows)
                lib_md5 = hashlib.md5(iq.tobytes()).hexdigest()
                out[name].update({"file_bytes": os.path.getsize(sink), "md5": h.hexdigest(), "md5_equals_library": h.hexdigest() == lib_md5})
                del iq
        finally:
            if sink != "/dev/null" and os.path.exists(sink):
                os.remove(sink)
    out["what"] = ("galileo-sdr-sim -e 20feb2022.rnx -l -6,51,100 -t 2022/02/20,12:00:00 -d %d -o <sink>: front-end + HIP + D2H + "
                   "write; value from the CLI's own 'Process time' line (as the reference prints it), wall time includes "
                   "HIP start-up and pinned allocations; md5_equals_library: the file against the same scenario through "
                   "gal_synth_run_host (untimed)" % seconds)
    return out


def leg_config(torch, pkg, workload, epochs, steps, local_rank, streams, flags=0, solo_launches=0):
    """A few steps of another BASELINE config at its real geometry (M-DYN = config 3, M-SYN24 = config 4 geometry with a
    bounded epoch count; "cboc" = the headline geometry in the opt-in CBOC mode; "syn12_4msps" = 12 SVs at 4 MS/s, a rate
    between the reference's and config 4's: window form 4; "syn12_4092ksps" = 12 SVs at 4.092 MS/s, thresholds that coincide), pipelined like the headline, on the
    headline's two streams (new streams would get whatever hardware queues are left: DESIGN.md section 6)."""
    n_samp, rate, n_slots, n_chan = 260000, 2.6e6, 16, 12
    if workload in ("syn24", "syn24_full"):
        n_samp, rate, n_slots, n_chan = 2500000, 25e6, 24, 24
    if workload == "syn12_4msps":
        n_samp, rate = 400000, 4.0e6
    if workload in ("syn12_4092ksps", "syn12_4092ksps_exact"):  # exactly two samples per half chip: the pattern thresholds coincide
        n_samp, rate = 409200, 4.092e6
        if workload.endswith("_exact"):
            flags |= pkg.synth.GAL_CFG_EXACT_REPLAY
    params = pkg.shard.rank_workload(0, epochs, n_chan=n_chan, n_slots=n_slots, samples_per_epoch=n_samp, sample_rate=rate,
                                     dyn_track=(workload == "dyn"))
    engines, outs = [], []
    while len(streams) < 2:  # (--pipeline 1)
        streams.append(torch.cuda.Stream())
    for _ in range(2):
        eng = pkg.SynthEngine(sample_rate=rate, samples_per_epoch=n_samp, n_slots=n_slots, device=local_rank,
                              flags=(pkg.synth.GAL_CFG_CBOC if workload == "cboc" else 0) | flags)
        eng.set_stream(streams[len(engines)].cuda_stream)
        eng.plan(params)
        engines.append(eng)
        outs.append(torch.empty(epochs * n_samp * 2, dtype=torch.int16, device="cuda"))

    def run(n):
        stats = []
        for k in range(n + 2):
            j = k % 2
            if k >= 2:
                stats.append(engines[j].finish()[1])
            if k < n:
                engines[j].execute(outs[j].data_ptr())
        return stats

    run(2 if workload == "syn24_full" else 8)  # (the legs in front of this one leave the device idle: CLI sink, allocations)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    stats = run(steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert all(x["chain_mismatch"] == 0 for x in stats)
    solo = None
    if solo_launches:  # the kernel with ONE handle in flight (what `roofline.frac` is for the headline)
        one = []
        for _ in range(solo_launches):
            engines[0].execute(outs[0].data_ptr())
            one.append(engines[0].finish()[1]["ms_synth"])
        solo = sum(one) / len(one)
    for e in engines:
        e.close()
    del outs[:]
    torch.cuda.empty_cache()
    value = epochs * n_samp * steps / dt / 1e6
    return {"value": round(value, 1), "unit": "Msamples/s", "x_realtime": round(value * 1e6 / rate, 1), "ms_per_step": round(dt / steps * 1e3, 3),
            "epochs": epochs, "channels": n_chan, "samples_per_epoch": n_samp, "steps": steps,
            "avg_kernel_ms": round(sum(x["ms_synth"] for x in stats) / len(stats), 3), "window_mode": stats[-1].get("window_mode"),
            **({"kernel_one_handle_ms": round(solo, 4)} if solo is not None else {})}


def leg_fresh_plan(torch, pkg, engines, outs, n_samp, rate, n_slots, n_chan, epochs, steps, resident_ms, dyn_track=False):
    """the engine on FRESH parameters.  The headline re-executes one resident plan; no caller runs the same 120 s
    twice -- the reference computes its parameters between epochs (src/galileo-sdr.cpp:450-479).  Here every step gets a scenario of
    its own (another seed: other Dopplers, code phases, pages) and is plan + execute + finish, on the headline's two handles and one
    host thread:  finish(j);  execute(j) -- which makes the plan that was STAGED while j's batch before was in flight the handle's,
    enqueues its upload and the walkers behind it --;  gal_synth_plan_async(j, the scenario after next) with the batch just executed
    in flight: validation, lists and the SoA split into pinned memory run under the two batches on the device -- plan(k+1) under
    execute(k).  The parameter sets are made before the timed region (producing them is the front-end's job, row f1).  The outputs
    of the last two steps -- two different seeds -- are then compared with the oracle, every int16 of every epoch (by cpu_baseline,
    which gets them as `pending`)."""
    import hashlib

    depth = len(engines)
    warm = 2 * depth
    n_sets = min(steps + warm, 64)  # (beyond 64 steps the seeds repeat, 64 steps apart: 3.4 MB of records each)
    sets = [pkg.shard.rank_workload(1000 + k, epochs, n_chan=n_chan, n_slots=n_slots, samples_per_epoch=n_samp, sample_rate=rate, dyn_track=dyn_track)
            for k in range(n_sets)]
    out2 = [torch.empty_like(outs[0]) for _ in engines]  # (its own outputs: the headline's stay as they are for the oracle's verdict on them)
    which = [None] * depth
    host = {"finish": 0.0, "plan": 0.0, "execute": 0.0}
    stamps = []  # host time at which each step's finish returned

    staged = [None] * depth

    def run(first, n):
        """n steps, each with ONE plan call: the handles come in with the first `depth` scenarios staged (as a caller in steady state
        always has the next batch planned) and leave with the scenarios of the steps behind the last one staged."""
        stats = []
        inflight = [False] * depth
        for k in range(n):
            j = k % depth
            ta = time.perf_counter()
            if inflight[j]:
                stats.append(engines[j].finish()[1])
                stamps.append(time.perf_counter())
            tb = time.perf_counter()
            if staged[j] is None:  # (pipeline fill: nothing was planned ahead for this handle)
                staged[j] = (first + k) % n_sets
                engines[j].plan(sets[staged[j]], wait=False)
            tc = time.perf_counter()
            engines[j].execute(out2[j].data_ptr())
            which[j], staged[j] = staged[j], None
            inflight[j] = True
            td = time.perf_counter()
            staged[j] = (first + k + depth) % n_sets  # the scenario this handle runs next, planned while its batch is in flight
            engines[j].plan(sets[staged[j]], wait=False)
            te = time.perf_counter()
            host["finish"] += tb - ta
            host["execute"] += td - tc
            host["plan"] += (tc - tb) + (te - td)
        for k in range(n, n + depth):
            j = k % depth
            if inflight[j]:
                stats.append(engines[j].finish()[1])
                stamps.append(time.perf_counter())
                inflight[j] = False
        return stats

    run(0, warm)
    torch.cuda.synchronize()
    host.update(finish=0.0, plan=0.0, execute=0.0)
    del stamps[:]
    t0 = time.perf_counter()
    stats = run(warm, steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert len(stats) == steps and all(x["chain_mismatch"] == 0 for x in stats)
    ms = dt / steps * 1e3
    # finish-to-finish intervals by the carrier passes the step needed (a scenario in which a channel's Doppler passes through zero
    # has carrier cycles of millions of samples; some of its legs are walked again in a second pass)
    by_passes = {}
    for k in range(1, len(stats)):
        by_passes.setdefault(int(stats[k]["walk_passes"]), []).append((stamps[k] - stamps[k - 1]) * 1e3)
    # what the oracle is to look at (cpu_baseline does, the only place of this file that touches the checker)
    pending = [({"seed": 1000 + which[j], "params_md5": hashlib.md5(sets[which[j]].tobytes()).hexdigest()[:12]}, sets[which[j]], out2[j])
               for j in range(depth)]
    return {"value": round(epochs * n_samp * steps / dt / 1e6, 1), "unit": "Msamples/s", "ms_per_step": round(ms, 4), "steps": steps,
            "distinct_parameter_sets": n_sets, "handles": depth,
            "plan_ms": round(sum(x["ms_plan"] for x in stats) / len(stats), 4), "h2d_ms": round(sum(x["ms_h2d"] for x in stats) / len(stats), 4),
            "ratio_to_resident_plan_step": round(ms / resident_ms, 4) if resident_ms else None,
            "steps_with_a_repeated_synthesis": sum(x.get("synth_runs", 1) > 1 for x in stats), "walk_passes_max": max(x["walk_passes"] for x in stats),
            "avg_kernel_ms": round(sum(x["ms_synth"] for x in stats) / len(stats), 4), "avg_walk_ms": round(sum(x["ms_walk"] for x in stats) / len(stats), 4),
            "host_ms_per_step": {k: round(v / steps * 1e3, 4) for k, v in host.items()},
            "ms_per_step_by_walk_passes": {str(k): {"steps": len(v), "ms": round(sum(v) / len(v), 4)} for k, v in sorted(by_passes.items())},
            "what": "step = gal_synth_plan_async (a new scenario: host validation + SoA split into pinned memory, staged while the handle's "
                    "batch before is in flight) + execute (commits the staged plan: upload enqueued, walkers behind it) + finish; %d handles, "
                    "one host thread; plan_ms = host time of the plan call, h2d_ms = device time of its upload, both per step" % depth}, pending


def profiled_kernel_ms(kind="bench"):
    """Average k_synth<12,false> duration in the newest committed rocprofv3 --kernel-trace --stats summary
    (not live: bench.py cannot run rocprofv3 on itself).  kind "bench": profiles/*_bench_kernel_stats.csv, this bench
    command as the driver runs it (two handles in flight, consecutive launches overlap); kind "standalone":
    profiles/*_standalone_kernel_stats.csv, `bench.py --pipeline 1 --no-extras --no-cpu-baseline` (one handle: every
    launch runs alone, so this average IS a per-launch cost and can serve as a roofline denominator)."""
    import csv
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_%s_kernel_stats.csv" % kind)))
    if not files:
        return None, None
    try:
        for r in csv.DictReader(open(files[-1])):
            if "k_synth" in r["Name"] and "12" in r["Name"] and "k_synth_g" in r["Name"]:
                return round(float(r["AverageNs"]) / 1e6, 4), os.path.basename(files[-1])
    except Exception:
        pass
    return None, None


def kernel_source_sha256():
    """Hash of the sources the profiled kernel is built from (tools/pmc_synth.sh writes the same into its summary)."""
    import hashlib

    src = os.path.join(ROOT, "galileo-sdr-sim_amd", "csrc")
    return hashlib.sha256(b"".join(open(os.path.join(src, f), "rb").read() for f in ("synth_group.hip", "synth_common.h", "synth_dev.h"))).hexdigest()


def profile_is_current(path):
    """A committed profile summary belongs to THIS build of the kernel (a stale file would go unnoticed)."""
    try:
        return json.load(open(path)).get("kernel_source_sha256") == kernel_source_sha256()
    except Exception:
        return False


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
        if not profile_is_current(files[-1]):  # measured on another build of the kernel: not this line's traffic
            return None, os.path.basename(files[-1]) + " (STALE: the kernel's sources have changed since it was measured)"
        return int(d["hbm_bytes_per_launch"]), os.path.basename(files[-1])
    except Exception:
        return None, None


def measured_issue(kernel_ms, channel_samples):
    """What actually bounds k_synth_g: instruction issue, not HBM.  From the newest committed PMC summary
    (profiles/*_pmc_k_synth_all.json: SQ_INSTS_VALU, SQ_LDS_IDX_ACTIVE per launch): VALU wave-instructions x 4 cycles over the
    chip's 1024 SIMDs at the 2.4 GHz the device reports, and LDS array cycles over its 256 LDS units, against the LIVE kernel time.
    (tools/issue_ubench: every VALU instruction of this path costs 4.1-4.3 cycles of that clock back to back, i.e. the device
    runs a pure VALU stream at ~2.25 GHz; at that clock the fractions are 7 % higher.)"""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_k_synth_all.json")))
    if not files or not kernel_ms or not profile_is_current(files[-1]):
        return None
    try:
        d = json.load(open(files[-1]))
        valu = float(d["SQ_INSTS_VALU"])
        out = {"bound": "valu issue", "valu_wave_instructions": int(valu), "cycles_per_instruction": 4, "simds": 1024, "clock_ghz": 2.4,
               "valu_ms": round(valu * 4 / 1024 / 2.4e9 * 1e3, 4), "kernel_ms": round(kernel_ms, 4),
               "frac": round(valu * 4 / 1024 / 2.4e9 * 1e3 / kernel_ms, 4),
               "per_channel_sample": round(valu * 64 / channel_samples, 3),
               "source": os.path.basename(files[-1]), "is_live": False}
        if "SQ_LDS_IDX_ACTIVE" in d:
            lds = float(d["SQ_LDS_IDX_ACTIVE"])
            out["lds_array_cycles"] = int(lds)
            out["lds_frac"] = round(lds / 256 / 2.4e9 * 1e3 / kernel_ms, 4)
            if "SQ_LDS_BANK_CONFLICT" in d:
                out["lds_bank_conflict_cycles"] = int(float(d["SQ_LDS_BANK_CONFLICT"]))
        return out
    except Exception:
        return None


# A/B experiments only (tools/): GAL_BENCH_HOOKS=1 runs the GAL_TEST_HOOKS build of the library, whose environment
# switches (GAL_SYNTH_RW=0 ...) select code paths; such a line carries "hooks_build": true and is never a result.
HOOKS_BUILD = bool(os.environ.get("GAL_BENCH_HOOKS"))


def exchange_report(dist, backend, ctl, report):
    """The report of a run -- MAX of the time, SUM of samples and checksums, per-rank detail -- over RCCL (the default group)
    when the backend is nccl.  Should the communicator fail to come up on this node the measurement is not lost with it:
    the same reductions then go through the gloo group that carried the barriers, and the line says so
    ("report_backend").  Whether to fall back is AGREED over that gloo group (MIN of an ok flag) before any rank takes the
    other path: a rank-local decision would leave the ranks in different collective sequences.  report(group, device) does
    the exchange; returns its four values + the backend used."""
    if dist is None:
        return report(None, "cpu") + (None,)
    if backend != "nccl":
        return report(None, "cpu") + (backend,)
    res, err = None, None
    try:
        res = report(None, "cuda")
    except Exception as exc:  # noqa: BLE001 -- whatever RCCL raises, the gloo group is the way out
        err = exc
    if ctl["group"] is None:
        if err is not None:
            raise err
        return res + ("rccl",)
    import torch

    ok = torch.tensor([0 if err is not None else 1], dtype=torch.int32)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=ctl["group"])
    if int(ok.item()) == 1:
        return res + ("rccl",)
    why = type(err).__name__ if err is not None else "on another rank"
    sys.stderr.write("bench: report over RCCL failed (%s%s); every rank falls back to the gloo group\n" % (
        why, ": %s" % err if err is not None else ""))
    return report(ctl["group"], "cpu") + ("gloo (RCCL failed: %s)" % why,)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100, help="timed steps (1.6 ms each; the pipeline is empty at both "
                    "ends of the timed region, so few steps mostly measure its fill and drain)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--epochs", type=int, default=1199, help="epochs per step (default: the 120 s scenario)")
    ap.add_argument("--channels", type=int, default=12)
    ap.add_argument("--chunk", type=int, default=0, help="samples per lane (0 = auto)")
    ap.add_argument("--workload", default="syn12", choices=["syn12", "syn24", "dyn", "locations"],
                    help="syn12 = headline M-SYN12 (BASELINE configs[1] size); syn24 = config 4 geometry (24 SVs, "
                    "25 MS/s); dyn = config 3 (12 SVs, Doppler from a 10 Hz circular track); locations = config 5 "
                    "literally: rank r simulates static site r of 8 for 300 s from tests/golden/20feb2022.rnx through "
                    "the real host front-end (7-10 SVs per site, so the ranks' work differs)")
    ap.add_argument("--as-rank", default=None, metavar="R/N", help="--shard scenario: do the work of rank R of N in THIS process, alone on "
                    "the device (no process group): its epoch range, walker and kernel time.  tools/strong_split_alone.sh runs every rank "
                    "of a world this way, one after the other, to read shard.epoch_range's balance off ONE GPU without the ranks "
                    "disturbing each other; says nothing about scaling")
    ap.add_argument("--site", type=int, default=None, help="--workload locations: the site (0..7 of shard.LOCATIONS) this process "
                    "simulates instead of site `rank` -- the per-site 1-GPU baseline of config 5 (config5_sites.sh (a tool of rounds 3-5: git history))")
    ap.add_argument("--pipeline", type=int, default=2, choices=[1, 2, 3, 4],
                    help="engine handles in flight: 2 = software pipeline, the NCO walk of step k+1 (latency "
                    "bound, on the handle's high-priority stream) runs beside the synthesis kernel of step k (issue "
                    "bound); every step still does the complete pass into its own buffers (measured: 1 / 2 / 3 "
                    "handles = 2.19 / 1.67 / 1.85 ms per step)")
    ap.add_argument("--shard", default="scenarios", choices=["scenarios", "scenario"],
                    help="N > 1: 'scenarios' = one independent scenario per rank (weak scaling, the default and the "
                    "headline); 'scenario' = ONE scenario cut into contiguous epoch ranges, every rank walks the whole "
                    "NCO chain and synthesises its own range (strong scaling, no exchange)")
    ap.add_argument("--signal", default="boc11", choices=["boc11", "cboc"], help="boc11 = the reference's signal (headline); "
                    "cboc = the opt-in CBOC(6,1,1/11) mode (GAL_CFG_CBOC)")
    ap.add_argument("--preroll-ms", type=float, default=300.0,
                    help="device wake-up in front of the W warm-up steps: un-timed passes of the same step for this long "
                    "(reported as preroll_steps).  A device that comes out of idle runs its first ~100 ms below its "
                    "sustained clocks, and W = 5 steps are 6 ms of work; 0 = none")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fresh-plan", action="store_true", help="skip configs.fresh_plan (the same step with a new scenario planned every step)")
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed-for-headline legs (kernel + D2H, CLI file "
                    "sink, M-DYN, M-SYN24) that the default 1-GPU run reports under e2e / configs")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the synthesis engine has no CPU fallback")
    # GAL_BENCH_DEVICE / GAL_BENCH_BACKEND: rehearsal of the N > 1 launch path on a box with ONE GPU (every rank on the
    # same device, gloo instead of RCCL for the barrier and the report reductions) -- exercises rank_workload /
    # epoch_range / reduce_report / gal_synth_execute_range together on real hardware; says nothing about scaling
    if os.environ.get("GAL_BENCH_DEVICE"):
        local_rank = int(os.environ["GAL_BENCH_DEVICE"])
    backend = os.environ.get("GAL_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dist = None

    ctl = {"group": None}  # the CPU-side (gloo) group that carries the barriers, see below

    def init_process_group():
        # GAL_BENCH_FORCE_DIST=1: initialise the process group for world size 1 as well (launched by torch.distributed.run
        # --nproc-per-node 1): the RCCL calls of the N > 1 path -- init, barrier, the two all_reduce of the report,
        # all_gather_object -- on a box with one GPU (rccl_one_rank.sh (a tool of rounds 3-5: git history))
        if not (world > 1 or os.environ.get("GAL_BENCH_FORCE_DIST")):
            return None
        import torch.distributed as dist_mod

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            # RCCL carries what the path has to exchange -- the report reductions behind the timed region.  The BARRIERS
            # around the timed region go through a gloo group of the same ranks (process barrier on the CPU, then
            # torch.cuda.synchronize()): an RCCL barrier is a device kernel on one more hardware queue, and the first half
            # dozen steps behind it run 20-30 % slow (profiles/archive/r03k_rccl_slowdown6.log).  GAL_BENCH_RCCL=eager creates the
            # communicator here instead of at its first collective, GAL_BENCH_BARRIER=rccl sends the barriers through it.
            if os.environ.get("GAL_BENCH_RCCL", "lazy") == "eager":
                dist_mod.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
            else:
                dist_mod.init_process_group(backend="nccl")
            if os.environ.get("GAL_BENCH_BARRIER", "gloo") == "gloo":
                ctl["group"] = dist_mod.new_group(backend="gloo")
        else:
            dist_mod.init_process_group(backend=backend)
        return dist_mod

    # When the process group is initialised matters on this runtime: HIP creates the hardware queue of a stream at the
    # stream's first use, and which queues end up next to each other decides how well the walker chain of one handle runs
    # beside the synthesis kernel of the other (profiles/archive/r03k_rccl_slowdown*.log, r03k_queue_map.log: the same bench reads
    # 1.25 ms per step in a plain process and 1.42 with the RCCL communicator created first).  "late" (default) = after
    # the engines exist and a first step has used every one of their streams; "early" = first thing, "mid" = after the
    # engines are planned but before their first step.
    pg_order = os.environ.get("GAL_BENCH_PG_ORDER", "late")
    if pg_order == "early":
        dist = init_process_group()

    from __graft_entry__ import load_pkg

    pkg = load_pkg()
    n_samp, rate, n_slots = 260000, 2.6e6, 16
    if args.workload == "syn24":
        n_samp, rate, n_slots = 2500000, 25e6, 24
        args.channels = 24
    # each rank: an independent scenario of identical size (different seed)
    strong = args.shard == "scenario"
    site = None
    if args.workload == "locations":
        nav = os.path.join(ROOT, "tests", "golden", "20feb2022.rnx")
        params, site = pkg.shard.rank_location_scenario(pkg.Scenario, nav, args.site if args.site is not None else (0 if strong else rank))
        args.epochs = params.shape[0]
        args.channels = int((params["prn"] > 0).sum(axis=1).max())
    else:
        params = pkg.shard.rank_workload(0 if strong else rank, args.epochs, n_chan=args.channels, n_slots=n_slots,
                                         samples_per_epoch=n_samp, sample_rate=rate, dyn_track=(args.workload == "dyn"))
    e_first, e_count = pkg.shard.epoch_range(rank, world, args.epochs) if strong else (0, args.epochs)
    as_rank = None
    if args.as_rank:
        if not strong or world != 1:
            raise SystemExit("--as-rank needs --shard scenario in a single process")
        as_rank = tuple(int(v) for v in args.as_rank.split("/"))
        e_first, e_count = pkg.shard.epoch_range(as_rank[0], as_rank[1], args.epochs)
    depth = args.pipeline
    engines, outs, streams = [], [], []
    for k in range(depth):
        eng = pkg.SynthEngine(sample_rate=rate, samples_per_epoch=n_samp, n_slots=n_slots, device=local_rank,
                              chunk_samples=args.chunk, flags=pkg.synth.GAL_CFG_CBOC if args.signal == "cboc" else 0,
                              test_hooks=HOOKS_BUILD)
        st = torch.cuda.Stream()
        eng.set_stream(st.cuda_stream)
        eng.plan(params)  # inputs resident in HBM before the timed region
        engines.append(eng)
        streams.append(st)
        outs.append(torch.empty(e_count * n_samp * 2, dtype=torch.int16, device="cuda"))
    out = outs[0]
    if pg_order == "mid":
        dist = init_process_group()

    def barrier():
        if dist is not None:
            dist.barrier(group=ctl["group"])  # (None = the default group)
        torch.cuda.synchronize()

    def run(n_steps):
        """n_steps complete passes, `depth` of them in flight; returns the per-step stats."""
        all_stats = []
        inflight = [False] * depth
        for k in range(n_steps):
            j = k % depth
            if inflight[j]:
                all_stats.append(engines[j].finish()[1])
                if step_clock is not None:
                    step_clock.append(time.perf_counter())
            engines[j].execute(outs[j].data_ptr(), e_first, e_count)
            inflight[j] = True
        for k in range(n_steps, n_steps + depth):
            j = k % depth
            if inflight[j]:
                all_stats.append(engines[j].finish()[1])
                if step_clock is not None:
                    step_clock.append(time.perf_counter())
                inflight[j] = False
        return all_stats

    # GAL_BENCH_STEP_TIMES=1: host time at which every finish() of the timed region returned, summarised on stderr
    step_clock = None

    preroll_steps = 0
    if pg_order not in ("early", "mid"):
        # first use of every stream of every handle: their hardware queues exist from here on; then the process group; the
        # device wake-up comes after it (RCCL's set-up takes a second or two, in which the device would go idle again)
        run(depth)
        preroll_steps += depth
        dist = init_process_group()
    t_pre = time.perf_counter()
    while (time.perf_counter() - t_pre) * 1e3 < args.preroll_ms:
        run(2 * depth)
        preroll_steps += 2 * depth
    run(args.warmup)
    barrier()
    t0 = time.perf_counter()
    if os.environ.get("GAL_BENCH_STEP_TIMES"):
        step_clock = [t0]
    step_stats = run(args.steps)
    t_run = time.perf_counter()
    barrier()
    elapsed = time.perf_counter() - t0
    if step_clock is not None:
        gaps = [(b - a) * 1e3 for a, b in zip(step_clock[:-1], step_clock[1:])]
        order = sorted(range(len(gaps)), key=lambda i: -gaps[i])[:6]
        sys.stderr.write("step times [ms]: first finish after %.3f, median gap %.3f, largest %s, end barrier %.3f, region %.3f\n" % (
            gaps[0], sorted(gaps)[len(gaps) // 2], ", ".join("#%d %.3f" % (i, gaps[i]) for i in order),
            (time.perf_counter() - t_run) * 1e3, elapsed * 1e3))
        step_clock = None
    assert len(step_stats) == args.steps
    stats = step_stats[-1]
    ms_synth = sum(s["ms_synth"] for s in step_stats)
    ms_walk = sum(s["ms_walk"] for s in step_stats)
    assert all(s["chain_mismatch"] == 0 for s in step_stats)
    # outside the timed region: the same kernel without a co-running walker (one handle), for the roofline note
    solo_n = args.steps
    if depth > 1:
        inflight_stats = []
        solo_n = 10
        for _ in range(solo_n):
            engines[0].execute(outs[0].data_ptr(), e_first, e_count)
            inflight_stats.append(engines[0].finish()[1])
        solo_ms = sum(x["ms_synth"] for x in inflight_stats) / len(inflight_stats)
    else:
        solo_ms = ms_synth / args.steps
    samples_per_step = e_count * n_samp  # this rank's share
    # integrity of what was timed: a checksum of the last output (outside the timed region)
    chk = 0
    v32 = out.view(torch.int32)
    for a in range(0, v32.numel(), 1 << 28):  # in pieces: the int64 widening of a 60 GB output is 120 GB
        chk = (chk + int(v32[a:a + (1 << 28)].to(torch.int64).sum().item())) & 0xFFFFFFFF
    # The report: MAX of the time, SUM of samples and checksums, per-rank detail (exchange_report)
    local_elapsed, local_samples, local_chk = elapsed, samples_per_step * args.steps, chk

    def report(group, device):
        el, tot, ck = pkg.shard.reduce_report(dist, device, local_elapsed, local_samples, local_chk, group=group)
        pr = None
        if dist is not None:
            # per-rank detail (N > 1): the epoch range, the walker and kernel time of this rank's steps and the carrier legs it
            # walked -- in the strong split (--shard scenario) a rank walks the prefix [0, end of its range), not the whole plan
            mine = {"rank": rank, "epochs": [int(e_first), int(e_first + e_count)], "avg_walk_ms": round(ms_walk / args.steps, 4),
                    "avg_kernel_ms": round(ms_synth / args.steps, 4), "legs_walked": int(engines[0].walk_counts()[0])}
            pr = [None] * world
            dist.all_gather_object(pr, mine, group=group)
        return el, tot, ck, pr

    # (RCCL prints a version banner when a communicator comes up -- on stdout, which belongs to the ONE JSON line: file
    # descriptor 1 points at stderr while the report is exchanged)
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    try:
        elapsed, total_samples, chk, per_rank, report_backend = exchange_report(dist, backend, ctl, report)
    finally:
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)
    value = total_samples / elapsed / 1e6

    if rank == 0:
        avg_synth_ms = ms_synth / args.steps
        traffic, traffic_src = measured_traffic() if (args.epochs == 1199 and args.workload == "syn12" and args.channels == 12
                                                          and e_count == args.epochs) else (None, None)
        achieved = 4.0 * samples_per_step / (avg_synth_ms * 1e-3) / 1e9 if avg_synth_ms > 0 else 0.0
        step_ms = elapsed / args.steps * 1e3
        step_achieved = 4.0 * samples_per_step / (step_ms * 1e-3) / 1e9
        solo_achieved = 4.0 * samples_per_step / (solo_ms * 1e-3) / 1e9
        prof_ms, prof_src = profiled_kernel_ms("bench") if traffic is not None else (None, None)
        prof1_ms, prof1_src = profiled_kernel_ms("standalone") if traffic is not None else (None, None)
        line = {
            "metric": METRIC,
            "value": round(value, 3),
            "unit": "Msamples/s",
            "x_realtime": round(value / 2.6, 2),
            "n_gpus": world,
            **({"rehearsal": "all %d ranks on GPU %d, backend %s: launch-path check, NOT a scaling measurement" % (
                world, local_rank, backend)} if os.environ.get("GAL_BENCH_DEVICE") and world > 1 else {}),
            **({"sink": "none: every rank synthesises into its own HBM (the engine).  With a file sink behind it (galileo-sdr-sim --sites) each "
                        "rank is bounded by its device->host link and the host's page cache, 1.9-2.3 G samples/s per process on this "
                        "host (DESIGN.md sections 6 and 7), and an aggregate over ranks measures the host, not the engine"} if world > 1 else {}),
            **({"ranks": per_rank, "report_backend": report_backend,
                # walker chain + synthesis per step, slowest rank over the mean: what the split leaves on the table (strong split:
                # shard.epoch_range cuts the ranges so that walk(prefix + range) + synth(range) is level)
                "rank_imbalance": round(max(r["avg_walk_ms"] + r["avg_kernel_ms"] for r in per_rank) /
                                        max(1e-9, sum(r["avg_walk_ms"] + r["avg_kernel_ms"] for r in per_rank) / len(per_rank)), 3)}
               if per_rank else {}),
            **({"as_rank": {"rank": as_rank[0], "of": as_rank[1], "epochs": [int(e_first), int(e_first + e_count)],
                            "avg_walk_ms": round(ms_walk / args.steps, 4), "avg_kernel_ms": round(ms_synth / args.steps, 4),
                            "legs_walked": int(engines[0].walk_counts()[0]),
                            "note": "this rank's share of ONE scenario, run alone on the device: a balance check, not a scaling measurement"}}
               if as_rank else {}),
            "steps": args.steps,
            "warmup": args.warmup,
            "preroll_steps": preroll_steps,  # un-timed device wake-up in front of the warm-up steps (--preroll-ms)
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "f64 (NCO phases) + int16x2 (packed accumulate) -> int16 IQ",
            "data": "synthetic" if args.workload != "locations" else "broadcast ephemerides of tests/golden/20feb2022.rnx through the host front-end (no measured IQ exists for this path)",
            "config": {
                "workload": {"syn12": "M-SYN12: static-geometry 12-SV E1B/C", "syn24": "M-SYN24: 24-SV E1B/C",
                             "dyn": "M-DYN: 12-SV E1B/C, 10 Hz circular user motion",
                             "locations": "config 5: static site per rank from 20feb2022.rnx (%s: %s, %d SVs), 300 s"
                             % ("site %d" % args.site if args.site is not None else "rank 0", site, args.channels)}[args.workload]
                + (", ONE scenario of %d epochs x %d samples @%.1f MS/s cut into epoch ranges over the ranks" if strong
                   else ", %d epochs x %d samples @%.1f MS/s per GPU (one independent scenario per rank)") % (
                    args.epochs, n_samp, rate / 1e6),
                "channels": args.channels,
                "chunk_samples": stats["chunk_samples"],
                "walk_passes": stats["walk_passes"], "synth_runs_max": max(s.get("synth_runs", 1) for s in step_stats),
                "chain_mismatch": stats["chain_mismatch"],
                "pipeline_depth": depth, **({"hooks_build": True} if HOOKS_BUILD else {}),
                **({"variant_lib": os.environ["GAL_SYNTH_LIB"]} if os.environ.get("GAL_SYNTH_LIB") else {}),
                "window_mode": stats.get("window_mode"),  # 1: resampled-window chips (galsynth.h)
                "kernel_family": stats.get("kernel_family"),  # 1: k_synth_g + k_repair_g (galsynth.h: gal_synth_stats_t)
                "output_checksum": "%08x" % chk,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": ("k_synth_g<%d,false>%s" % (min(args.channels, 12), " (+ accumulate launch)" if args.channels > 12 else "")
                           if stats.get("kernel_family") == 1 else
                           "k_synth<%d,false,%d,%d>%s" % (min(args.channels, 12), 1 if args.signal == "cboc" else 0,
                                                         (stats.get("window_mode") or 0) & 15,
                                                         " (+ accumulate launch)" if args.channels > 12 else "")),
                # family 1 (synth_group.hip): behind k_synth_g, k_repair_g replays the groups it could not decide -- a second,
                # small kernel of the same pass; its time is in every step and in `sustained`, and reported here
                **({"repair": {"kernel": "k_repair_g", "avg_ms": round(sum(x.get("ms_repair", 0.0) for x in step_stats) / len(step_stats), 4),
                               "groups_per_launch": stats.get("repaired_groups"),
                               "of_groups": int(samples_per_step // 16)}} if stats.get("kernel_family") == 1 else {}),
                # achieved / frac: algorithmic bytes per launch / the kernel's launch duration with the kernel running ALONE
                # (HIP events on its stream, one handle, measured live right behind the timed region).  Inside the timed
                # region two handles are in flight and consecutive launches OVERLAP -- the tail of one runs beside the head
                # of the next -- so the per-launch intervals there (`overlapped`, which is also what rocprofv3 reports for
                # this command) add up to more than the wall time and are not a per-launch cost; `sustained` is launches x
                # bytes over the timed region.  The tracked rocprofv3 summary of the kernel alone
                # (profiles/*_standalone_kernel_stats.csv: `bench.py --pipeline 1`) reproduces `frac`.
                "achieved": round(solo_achieved, 2),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(solo_achieved / HBM_PEAK_GBS, 5),
                "frac_uses": "algorithmic bytes per launch / avg_kernel_ms (HIP events on the kernel's stream, kernel alone: "
                             "%d launches of one handle right behind the timed region)" % solo_n,
                "traffic": traffic,
                "traffic_source": traffic_src,
                "traffic_is_live": False,
                "avg_kernel_ms": round(solo_ms, 4),
                "rocprof_avg_kernel_ms": prof1_ms,
                "rocprof_source": prof1_src,
                "rocprof_is_live": False,
                "frac_rocprof_standalone": round(4.0 * samples_per_step / (prof1_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if prof1_ms else None,
                "sustained": {"ms_per_launch": round(step_ms, 4), "achieved": round(step_achieved, 2),
                              "frac": round(step_achieved / HBM_PEAK_GBS, 5),
                              "what": "launches x algorithmic bytes over the timed region (= value x 4 B)"},
                "overlapped": {"avg_kernel_ms": round(avg_synth_ms, 4), "achieved": round(achieved, 2),
                               "frac": round(achieved / HBM_PEAK_GBS, 5), "rocprof_avg_kernel_ms": prof_ms, "rocprof_source": prof_src,
                               "what": "per-launch HIP-event intervals INSIDE the timed region (%d handles in flight)" % depth},
                "avg_walk_ms": round(ms_walk / args.steps, 4),
                "algorithmic_bytes_per_launch": 4 * samples_per_step,
                # the bound that binds: issue slots (north_star's HBM-write fraction above stays the headline `frac`)
                **({"issue": measured_issue(solo_ms, float(samples_per_step) * args.channels)}
                   if traffic is not None and measured_issue(solo_ms, float(samples_per_step) * args.channels) else {}),
            },
        }
        default_run = (world == 1 and args.workload == "syn12" and args.epochs == 1199 and args.channels == 12 and not strong
                       and args.signal == "boc11")
        if args.signal == "cboc":
            line["config"]["signal"] = "CBOC(6,1,1/11), opt-in mode (not the reference's signal, not the headline)"
            args.no_cpu_baseline = True
        pending = []
        if world == 1 and args.workload in ("syn12", "dyn") and not strong and args.signal == "boc11" and not args.no_fresh_plan:
            # the same step on FRESH parameters (plan + execute + finish, another scenario every step), on the headline's handles
            fp, pending = leg_fresh_plan(torch, pkg, engines, outs, n_samp, rate, n_slots, args.channels, args.epochs,
                                         args.steps, elapsed / args.steps * 1e3, dyn_track=(args.workload == "dyn"))
            line["configs"] = {"fresh_plan": fp}
            line["config"]["plan_ms"] = fp["plan_ms"]  # host time of one gal_synth_plan of this workload
        if world == 1 and not args.no_cpu_baseline:
            # (in front of the extras, which free the timed outputs: the checker's samples are compared with them)
            line["cpu_baseline"] = cpu_baseline(pkg, params, n_samp, rate, gpu_out=out, extra_checks=pending)  # (world 1: the whole scenario)
            v = line["cpu_baseline"].pop("output_vs_oracle", None)
            xc = line["cpu_baseline"].pop("extra_checks", None)
            if xc is not None:
                line["configs"]["fresh_plan"]["oracle_checks"] = xc
                line["configs"]["fresh_plan"]["output_equals_oracle"] = all(c["int16_different"] == 0 for c in xc)
            if v is not None:
                line["config"]["output_equals_oracle"] = v["equal"]
                line["config"]["output_vs_oracle"] = v
            if (v is not None and not v["equal"]) or (xc is not None and not line["configs"]["fresh_plan"]["output_equals_oracle"]):
                print(json.dumps(line))
                raise SystemExit("bench: an output differs from the oracle (headline: %s; fresh-plan scenarios: %s)" % (
                    v and v["int16_different"], xc and [c["int16_different"] for c in xc]))
        del pending
        if default_run and not args.no_extras:
            # untimed-for-headline legs (kernel-only above, kernel + D2H and the file sink here; configs 3/4)
            line["e2e"] = {"kernel_plus_d2h": leg_kernel_plus_d2h(torch, engines, outs, streams, e_first, e_count, n_samp)}
            for eng in engines:
                eng.close()
            del outs[:], out
            torch.cuda.empty_cache()
            line["e2e"]["file_sink"] = leg_file_sink(pkg)
            line["configs"] = {**line.get("configs", {}),
                               "dyn": leg_config(torch, pkg, "dyn", 2999, 24, local_rank, streams),
                               "syn24": leg_config(torch, pkg, "syn24", 600, 8, local_rank, streams),
                               # BASELINE config 4 at its FULL size: 600 s x 25 MS/s x 24 SVs = 15.0 G samples, 60 GB of IQ
                               # per handle kept in HBM (sample indices beyond 2^32)
                               "syn24_full": leg_config(torch, pkg, "syn24_full", 5999, 3, local_rank, streams),
                               # the opt-in CBOC(6,1,1/11) mode on the headline geometry (not the reference's signal)
                               "cboc": leg_config(torch, pkg, "cboc", 1199, 30, local_rank, streams),
                               # a sample rate between the window forms of rounds 2-4 (2.77 .. 7.7 MS/s): form 4
                               "syn12_4msps": leg_config(torch, pkg, "syn12_4msps", 1199, 20, local_rank, streams),
                               # a rate whose pattern thresholds crowd (4 x 1.023 MHz): k_synth_g's bisection instances (round 6; the
                               # exact-replay kernel before)
                               "syn12_4092ksps": leg_config(torch, pkg, "syn12_4092ksps", 1199, 20, local_rank, streams)}
            # the headline with the opt-in SAMPLED self-check (round 5's default: a rotating eighth of the leg positions per batch instead
            # of every leg of both chains in every batch), same run, same box: what full verification costs the step
            vs = leg_config(torch, pkg, "syn12", 1199, max(args.steps, 20), local_rank, streams, flags=pkg.synth.GAL_CFG_VERIFY_SAMPLED, solo_launches=10)
            line["roofline"]["verify_sampled"] = {"ms_per_step": vs["ms_per_step"], "value": vs["value"], "unit": "Msamples/s",
                                                  "steps": vs["steps"], "avg_kernel_ms": vs["avg_kernel_ms"],
                                                  # k_synth_g with one handle in flight and only the rotation's eighth of the verification
                                                  # beside it: the kernel itself has not changed since round 5 (BENCH_r05: 0.8413 ms = 18.5 %)
                                                  "kernel_one_handle_ms": vs["kernel_one_handle_ms"],
                                                  "kernel_one_handle_frac": round(4.0 * 1199 * 260000 / (vs["kernel_one_handle_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                                                  "what": "GAL_CFG_VERIFY_SAMPLED: every (epoch, leg) position re-walked once per 8 batches; "
                                                          "the headline re-walks every leg of both chains in every batch"}
        line["x_realtime"] = round(value * 1e6 / rate, 2)
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

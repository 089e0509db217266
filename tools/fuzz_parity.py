#!/usr/bin/env python3
"""Randomised parity soak on the GPU: HIP path vs the oracle, bit for bit, over randomly shaped batches
(slots, channels, epoch length, sample rate, chunking, Doppler incl. tiny / zero / sign flips / few-bit steps,
channels appearing, vanishing and being re-allocated, symbol counters near the page flip, code phases near the
wrap).  Test infrastructure: uses the oracle as the checker.   python tools/fuzz_parity.py [n_cases] [seed] [big]
GAL_FUZZ_HOOKS=1 runs the GAL_TEST_HOOKS build (e.g. with GAL_SCAN_BLOCK_LEGS=8: many blocks per slot in the carrier stitch of every batch, i.e. its look-back).
GAL_FUZZ_CBOC=1 runs the opt-in CBOC(6,1,1/11) mode against the checker's CBOC loop.
GAL_FUZZ_ASYNC=1 runs every case through gal_synth_plan_async / execute / finish on device buffers; a case that is cut in two STAGES its
second part while the first is in flight (plan(k+1) under execute(k) on one handle: the staged plan is committed by the next execute).
GAL_G_WIDE=1 with GAL_FUZZ_HOOKS=1: k_synth_g's wide instances (13-24 channels in one launch) for every batch that has more than 12.
GAL_FUZZ_GROUP=1 draws batches the default kernel of the reference geometry (k_synth_g + k_repair_g) can take, and reports how
many it took and how many 16-sample groups were replayed exactly."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401,E402  (initialise HIP through torch first)
from __graft_entry__ import load_pkg  # noqa: E402
from oracle_binding import oracle_run  # noqa: E402
from fuzz_cases import random_case as _random_case  # noqa: E402

pkg = load_pkg()
CBOC = bool(os.environ.get("GAL_FUZZ_CBOC"))


GROUP = bool(os.environ.get("GAL_FUZZ_GROUP"))


def random_case(rng, big=False):
    return _random_case(pkg, rng, big, group=GROUP)


ASYNC = bool(os.environ.get("GAL_FUZZ_ASYNC"))


def run_async(eng, parts, n_samp):
    """plan_async / execute / finish over the parts of one run (device buffers).  While a part is in flight a copy of the FIRST part
    is staged on the handle (plan(k+1) under execute(k)); the real next part -- which needs the carried state -- replaces it behind
    the finish; the last staged copy is committed by one more execute and must give the first part's bits once more."""
    outs, st, agg = [], None, None
    bufs = [torch.empty(q.shape[0] * n_samp * 2, dtype=torch.int16, device="cuda") for q in parts]
    eng.plan(parts[0], None, wait=False)
    for k, q in enumerate(parts):
        eng.execute(bufs[k].data_ptr())
        eng.plan(parts[0], None, wait=False)  # staged beside the batch in flight
        st, stats = eng.finish()
        outs.append(bufs[k].cpu().numpy())
        if agg is None:
            agg = stats
        else:
            agg["chain_mismatch"] += stats["chain_mismatch"]
            agg["repaired_groups"] += stats["repaired_groups"]
        if k + 1 < len(parts):
            eng.plan(parts[k + 1], st, wait=False)  # replaces the staged copy
    again = torch.empty_like(bufs[0])
    eng.execute(again.data_ptr())  # commits the staged copy of the first part
    _, stats_again = eng.finish()
    if stats_again["chain_mismatch"] or not np.array_equal(again.cpu().numpy(), outs[0]):
        agg["chain_mismatch"] += 1000000  # (reported as a mismatch of the case)
    return np.concatenate(outs), st, agg


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    big = len(sys.argv) > 3 and sys.argv[3] == "big"
    rng = np.random.default_rng(seed)
    t0 = time.time()
    bad = 0
    fb_total = 0
    n_run = 0
    samples = 0
    rejected = {}
    fam1 = repaired = 0
    for c in range(n_cases):
        p, n_samp, rate, chunk = random_case(rng, big)
        try:
            with pkg.SynthEngine(sample_rate=rate, samples_per_epoch=n_samp, n_slots=p.shape[1], device=0,
                                 chunk_samples=chunk, test_hooks=bool(os.environ.get("GAL_FUZZ_HOOKS")),
                                 flags=pkg.synth.GAL_CFG_CBOC if CBOC else 0) as eng:
                cut = int(rng.integers(1, p.shape[0])) if (p.shape[0] > 1 and rng.random() < 0.4) else 0
                if ASYNC:
                    iq, st, stats = run_async(eng, [p[:cut], p[cut:]] if cut else [p], n_samp)
                    stats2 = {"repaired_groups": 0}
                elif cut:  # the same run in two calls, the channel state carried by the caller
                    iq1, st1, stats = eng.run_host(p[:cut])
                    fb_total += eng.walk_counts()[2]
                    iq2, st, stats2 = eng.run_host(p[cut:], st1)
                    iq = np.concatenate([iq1, iq2])
                    stats["chain_mismatch"] += stats2["chain_mismatch"]
                else:
                    iq, st, stats = eng.run_host(p)
                fb_total += eng.walk_counts()[2]
        except pkg.GalSynthError as ex:
            # the engine may reject what the oracle also rejects (e.g. f_code / fs outside the window)
            msg = str(ex)
            key = msg.split(":")[-1].strip()[:50]
            try:
                oracle_run(p, n_samp, rate, cboc=CBOC)
            except Exception:
                rejected["(oracle too) " + key] = rejected.get("(oracle too) " + key, 0) + 1
                continue
            if "f_code / sample_rate" in msg or "continues without" in msg or "bad phase" in msg:
                rejected[key] = rejected.get(key, 0) + 1
                continue
            print("case %d: engine rejected: %s" % (c, msg))
            bad += 1
            continue
        ref_iq, ref_st = oracle_run(p, n_samp, rate, cboc=CBOC)
        fam1 += stats["kernel_family"] == 1
        repaired += stats["repaired_groups"] + (stats2["repaired_groups"] if cut else 0)
        n_run += 1
        samples += p.shape[0] * n_samp
        act = ref_st["prn"] > 0
        ok = (np.array_equal(iq, ref_iq) and stats["chain_mismatch"] == 0 and np.array_equal(st["prn"], ref_st["prn"])
              and np.array_equal(st["carr_phase"][act].view(np.uint64), ref_st["carr_phase"][act].view(np.uint64))
              and np.array_equal(st["page"][act], ref_st["page"][act]))
        if not ok:
            bad += 1
            print("case %d MISMATCH: rate %.4g slots %d epochs %d samples %d chunk %d passes %d" % (
                c, rate, p.shape[1], p.shape[0], n_samp, chunk, stats["walk_passes"]))
            np.save(os.path.join(ROOT, "gpurun_out", "fuzz_fail_%d_%d.npy" % (seed, c)), p)
    print("fuzz: %d cases, %d compared (%.1f M samples), %d bad, %d fallbacks, %d by k_synth_g (%d groups replayed), rejected %s, %.1f s" % (
        n_cases, n_run, samples / 1e6, bad, fb_total, fam1, repaired, rejected, time.time() - t0))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Randomised parity soak on the GPU: HIP path vs the oracle, bit for bit, over randomly shaped batches
(slots, channels, epoch length, sample rate, chunking, Doppler incl. tiny / zero / sign flips / few-bit steps,
channels appearing, vanishing and being re-allocated, symbol counters near the page flip, code phases near the
wrap).  Test infrastructure: uses the oracle as the checker.   python tools/fuzz_parity.py [n_cases] [seed] [big]
GAL_FUZZ_HOOKS=1 runs the GAL_TEST_HOOKS build (e.g. with GAL_SCAN_SINGLE_LEGS=0: the long-batch stitcher on every batch).
GAL_FUZZ_CBOC=1 runs the opt-in CBOC(6,1,1/11) mode against the checker's CBOC loop."""
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

pkg = load_pkg()
RESTART = 1
CBOC = bool(os.environ.get("GAL_FUZZ_CBOC"))


def random_case(rng, big=False):
    rate = float(rng.choice([2.047e6, 2.0465e6, 2.3e6, 2.6e6, 2.6e6, 2.6e6, 2.75e6, 2.78e6, 4.0e6, 4.092e6, 7.7e6, 8e6, 10e6, 12.5e6, 15.4e6, 16e6, 25e6, 25e6, 40e6]))
    n_slots = int(rng.choice([4, 8, 16, 16, 24, 40, 64]))
    n_chan = int(rng.integers(1, n_slots + 1))
    n_ep = int(rng.integers(1, 7))
    n_samp = int(rng.choice([rng.integers(16, 3000), rng.integers(3000, 70000), int(rate / 10) if rate <= 4.1e6 else 40000]))
    if big:  # reference geometry, many epochs: legs, translation and the stitcher at work
        rate, n_samp = 2.6e6, 260000
        n_slots = 16
        n_chan = int(rng.integers(6, 17))
        n_ep = int(rng.integers(20, 81))
    span = float(rng.choice([5.0, 300.0, 3500.0, 5000.0]))
    p = pkg.workloads.make_synthetic(n_epochs=n_ep, n_chan=n_chan, n_slots=n_slots, samples_per_epoch=n_samp,
                                     sample_rate=rate, seed=int(rng.integers(1 << 30)), doppler_span=span,
                                     drift_hz_per_epoch=float(rng.choice([-0.05, 0.0, 3.0, -40.0])),
                                     prns=[int(x) for x in (rng.permutation(50)[:n_chan] + 1 if n_chan <= 50 else rng.integers(1, 51, n_chan))])
    for j in range(n_chan):
        r = rng.random()
        if r < 0.15:   # exactly zero or tiny Doppler in some epochs
            e = rng.integers(0, n_ep)
            p["f_carr"][e:, j] = rng.choice([0.0, 1e-7, -3e-5, 0.02])
        elif r < 0.3:  # sign flip
            e = rng.integers(0, n_ep)
            p["f_carr"][e:, j] = -p["f_carr"][e:, j]
        elif r < 0.45:  # few-bit steps (ties at the wrap)
            k = int(rng.choice([50, 52, 53, 54]))
            d = p["f_carr"][:, j] / rate
            p["f_carr"][:, j] = np.round(d * 2.0 ** k) / 2.0 ** k * rate
        p["f_code"][:, j] = 1.023e6 + p["f_carr"][:, j] * 0.0006493506493506494
        if rng.random() < 0.25:  # code steps with few significant bits: the tie binade of k_synth's group advance moves up
            k = int(rng.integers(1, 16))
            st = (p["f_code"][:, j] * (1.0 / rate)).astype(np.float64)
            m = st.view(np.uint64)
            m = (m >> np.uint64(k) << np.uint64(k)) | np.uint64(1 << k)
            p["f_code"][:, j] = m.view(np.float64) * rate  # (the product may miss the crafted step by an ulp: still few-bit-ish)
        if rng.random() < 0.3:
            p["ibit0"][0, j] = int(rng.choice([498, 499, 0]))
        if rng.random() < 0.3:
            p["code_phase0"][int(rng.integers(0, n_ep)), j] = float(rng.choice([4091.99, 4092.0 + 0.3, 6137.9, 0.0]))
        if rng.random() < 0.15 and n_ep > 1 and p["prn"][-1, j] > 0:  # re-acquired mid-run: fresh carrier and page
            e = int(rng.integers(1, n_ep))
            if p["prn"][e, j] > 0:
                p["flags"][e, j] = RESTART
                p["carr_phase0"][e, j] = rng.uniform(-0.999, 0.999)
                p["page_init"][e, j] = p["page_next"][(e + 1) % n_ep, j]
        if rng.random() < 0.2 and n_ep > 2:  # vanish
            e = int(rng.integers(1, n_ep))
            p[e:, j] = np.zeros((), dtype=p.dtype)
            if rng.random() < 0.5 and e + 1 < n_ep:  # and come back as another PRN with a fresh carrier
                q = pkg.workloads.make_synthetic(n_epochs=n_ep, n_chan=1, n_slots=1, samples_per_epoch=n_samp,
                                                 sample_rate=rate, seed=int(rng.integers(1 << 30)),
                                                 prns=[int(rng.integers(1, 51))])
                p[e + 1:, j] = q[e + 1:, 0]
                p["flags"][e + 1, j] = RESTART
                p["carr_phase0"][e + 1, j] = rng.uniform(-0.999, 0.999)
                p["page_init"][e + 1, j] = q["page_next"][0, 0]
    chunk = int(rng.choice([0, 0, 0, 4 * int(rng.integers(1, 400)), 16 * int(rng.integers(1, 100))]))
    return p, n_samp, rate, chunk


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
    for c in range(n_cases):
        p, n_samp, rate, chunk = random_case(rng, big)
        try:
            with pkg.SynthEngine(sample_rate=rate, samples_per_epoch=n_samp, n_slots=p.shape[1], device=0,
                                 chunk_samples=chunk, test_hooks=bool(os.environ.get("GAL_FUZZ_HOOKS")),
                                 flags=pkg.synth.GAL_CFG_CBOC if CBOC else 0) as eng:
                cut = int(rng.integers(1, p.shape[0])) if (p.shape[0] > 1 and rng.random() < 0.4) else 0
                if cut:  # the same run in two calls, the channel state carried by the caller
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
    print("fuzz: %d cases, %d compared (%.1f M samples), %d bad, %d fallbacks, rejected %s, %.1f s" % (
        n_cases, n_run, samples / 1e6, bad, fb_total, rejected, time.time() - t0))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()

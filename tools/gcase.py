#!/usr/bin/env python3
"""Reproduce one case of tests/test_parity_gpu.py::test_randomised_soak (seed 2024) and locate the mismatching samples."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa
from __graft_entry__ import load_pkg
from oracle_binding import oracle_run
from fuzz_cases import random_case
pkg = load_pkg()
seed, want = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
for c in range(want + 1):
    p, n_samp, rate, chunk = random_case(pkg, rng, big=(c % 20 == 19))
print("case", want, "rate", rate, "shape", p.shape, "n_samp", n_samp, "chunk", chunk)
for flags in (0, 4):
    with pkg.SynthEngine(sample_rate=rate, samples_per_epoch=n_samp, n_slots=p.shape[1], device=0, chunk_samples=chunk, flags=flags) as eng:
        iq, st, stats = eng.run_host(p)
    ref, rst = oracle_run(p, n_samp, rate)
    bad = np.flatnonzero(iq != ref)
    print("flags", flags, {k: stats[k] for k in ("kernel_family", "window_mode", "repaired_groups", "chunk_samples", "n_active_max")}, "bad int16:", bad.size)
    if bad.size:
        smp = np.unique(bad // 2)
        ep = smp // n_samp
        pos = smp % n_samp
        print("  samples bad:", smp.size, "epochs", np.unique(ep), "first positions", pos[:12], "last", pos[-5:])
        print("  chunk idx", np.unique(pos // 1024)[:20], "group-in-chunk", np.unique((pos % 1024) // 16)[:20])
        for e in np.unique(ep)[:3]:
            act = p["prn"][e] > 0
            print("  epoch", e, "active", int(act.sum()), "code_phase0", p["code_phase0"][e][act][:8], "f_carr", p["f_carr"][e][act][:6], "flags", p["flags"][e][act][:8], "ibit0", p["ibit0"][e][act][:8])

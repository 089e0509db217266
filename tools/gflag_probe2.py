#!/usr/bin/env python3
"""Isolate which channel of a heavy-listing batch causes the listings (GPU): run each channel alone."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa
from __graft_entry__ import load_pkg
from fuzz_cases import random_case
pkg = load_pkg()
rng = np.random.default_rng(401)
shown = 0
for c in range(400):
    p, n_samp, rate, chunk = random_case(pkg, rng, False, group=True)
    with pkg.SynthEngine(sample_rate=rate, samples_per_epoch=n_samp, n_slots=p.shape[1], device=0) as eng:
        iq, st, stats = eng.run_host(p)
        if stats["kernel_family"] != 1:
            continue
        groups = p.shape[0] * ((n_samp + 15) // 16)
        frac = stats["repaired_groups"] / groups
        if frac < 0.01 or n_samp < 10000:
            continue
        print("case %d frac %.4f rate %.3g shape %s n_samp %d" % (c, frac, rate, p.shape, n_samp))
        for j in range(p.shape[1]):
            if not (p["prn"][:, j] > 0).any():
                continue
            q = np.zeros_like(p)
            q[:, j] = p[:, j]
            iq, st, s1 = eng.run_host(q)
            fj = s1["repaired_groups"] / groups
            if fj > 1e-3:
                e = int(np.argmax(p["prn"][:, j] > 0))
                print("   slot %d alone: frac %.4f family %d  f_carr %s  carr_phase0 %r flags %s code_phase0 %s f_code-1.023e6 %s" % (
                    j, fj, s1["kernel_family"], p["f_carr"][:, j].tolist(), float(p["carr_phase0"][e, j]), p["flags"][:, j].tolist(),
                    p["code_phase0"][:, j].tolist(), (p["f_code"][:, j] - 1.023e6).tolist()))
        shown += 1
        if shown >= 4:
            break

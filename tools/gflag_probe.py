#!/usr/bin/env python3
"""Which batches of the group-kernel soak list many groups for the exact replay?  (GPU)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa
from __graft_entry__ import load_pkg
from fuzz_cases import random_case
pkg = load_pkg()
rng = np.random.default_rng(401)
rows = []
for c in range(400):
    p, n_samp, rate, chunk = random_case(pkg, rng, False, group=True)
    with pkg.SynthEngine(sample_rate=rate, samples_per_epoch=n_samp, n_slots=p.shape[1], device=0) as eng:
        iq, st, stats = eng.run_host(p)
    if stats["kernel_family"] != 1:
        continue
    groups = p.shape[0] * ((n_samp + 15) // 16)
    frac = stats["repaired_groups"] / groups
    act = p["prn"] > 0
    fmin = np.abs(p["f_carr"][act]).min() if act.any() else 0
    rows.append((frac, rate, p.shape, n_samp, int(act.sum(axis=1).max()), fmin))
rows.sort(reverse=True)
for r in rows[:25]:
    print("frac %.4f rate %.3g shape %s n_samp %d nact %d min|f_carr| %.3g" % r)
fr = np.array([r[0] for r in rows])
print("cases", len(rows), "median frac", np.median(fr), "mean", fr.mean(), "share of cases with frac > 1e-3:", (fr > 1e-3).mean())
for rate in sorted(set(r[1] for r in rows)):
    x = np.array([r[0] for r in rows if r[1] == rate])
    print("rate %.3g: n %d mean frac %.5f median %.6f" % (rate, len(x), x.mean(), np.median(x)))

#!/usr/bin/env python3
"""Search seeds of the fuzz generator for batches whose carrier chain is not complete after one walker pass (tests of
the adaptive number of enqueued passes; round 2, when translations took a pass of their own: after two):
python tools/find_three_pass_batch.py [n_cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: F401,E402
import fuzz_parity as fz  # noqa: E402

pkg = fz.pkg
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
found = 0
for c in range(n_cases):
    rng = np.random.default_rng([seed, c])
    p, n_samp, rate, chunk = fz.random_case(rng, False)
    try:
        with pkg.SynthEngine(sample_rate=rate, samples_per_epoch=n_samp, n_slots=p.shape[1], device=0, chunk_samples=chunk) as eng:
            iq, st, stats = eng.run_host(p)
    except pkg.GalSynthError:
        continue
    if stats["walk_passes"] >= 2:  # (round 3: the stitch translates on the spot, an ordinary batch needs ONE pass)
        found += 1
        print("case %d: passes %d rate %.4g slots %d epochs %d samples %d chunk %d" % (
            c, stats["walk_passes"], rate, p.shape[1], p.shape[0], n_samp, chunk))
        if found >= 8:
            break
print("found", found)

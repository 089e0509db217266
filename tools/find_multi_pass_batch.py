#!/usr/bin/env python3
"""Which small batches need MORE than one carrier pass (walk + stitch)?  Runs random cases of tests/fuzz_cases.py on the GPU and prints
those whose chain was not complete after the first pass -- the tests of the repair paths (tests/test_parity_gpu.py::_hard_batch) need
one.  tools/find_multi_pass_batch.py [n_cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_pkg  # noqa: E402
from fuzz_cases import random_case  # noqa: E402

pkg = load_pkg()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 6
found = 0
for c in range(n_cases):
    p, n_samp, rate, chunk = random_case(pkg, np.random.default_rng([seed, c]), False)
    with pkg.SynthEngine(sample_rate=rate, samples_per_epoch=n_samp, n_slots=p.shape[1], device=0, chunk_samples=chunk) as eng:
        iq, st, stats = eng.run_host(p)
        walked, translated, fb = eng.walk_counts()
    if stats["walk_passes"] >= 2:
        found += 1
        print("case [%d, %d]: passes %d, rate %g, n_samp %d, chunk %d, shape %s, family %d, walked %d translated %d" % (
            seed, c, stats["walk_passes"], rate, n_samp, chunk, p.shape, stats["kernel_family"], walked, translated), flush=True)
        if found >= 12:
            break
print("%d of %d cases needed more than one pass" % (found, c + 1))

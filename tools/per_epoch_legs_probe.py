#!/usr/bin/env python3
"""One-epoch calls (INTEGRATION.md option B) against the number of carrier-walk legs per epoch (hooks build: GAL_WALK_LEGS):
a one-epoch batch has 8 legs x 16 slots = 128 walking lanes, each 32 500 samples long -- shorter legs shorten the chain."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch  # noqa: E401,F401
from __graft_entry__ import load_pkg
pkg = load_pkg()
n = 260000
p = pkg.workloads.make_synthetic(n_epochs=300, n_chan=9, n_slots=16, samples_per_epoch=n, seed=3)
for legs in (8, 16, 25, 32, 64, 125):
    os.environ["GAL_WALK_LEGS"] = str(legs)
    with pkg.SynthEngine(samples_per_epoch=n, n_slots=16, device=0, test_hooks=True) as eng:
        out = torch.empty(n * 2, dtype=torch.int16, device="cuda")
        eng.plan(p[:1]); eng.execute(out.data_ptr()); st, _ = eng.finish()
        tf = 0.0; mw = 0.0; ms = 0.0
        for e in range(1, 201):
            q = p[e:e+1].copy(); q["flags"][0, :] = 0
            eng.plan(q, st); eng.execute(out.data_ptr())
            t2 = time.perf_counter(); st, stats = eng.finish(); tf += time.perf_counter() - t2
            mw += stats["ms_walk"]; ms += stats["ms_synth"]
            assert stats["chain_mismatch"] == 0
        print("legs %3d: finish(wait) %.3f ms  ms_walk %.3f  ms_synth %.3f  walk_passes %d" % (legs, tf / 200 * 1e3, mw / 200, ms / 200, stats["walk_passes"]))

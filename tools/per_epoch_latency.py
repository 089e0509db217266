#!/usr/bin/env python3
"""Latency of the UNBATCHED integration (INTEGRATION.md option B as sketched: one gal_synth_run_host call per
0.1 s epoch, host buffers, state carried by the caller).  python tools/per_epoch_latency.py"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from __graft_entry__ import load_pkg
pkg = load_pkg()
p = pkg.workloads.make_synthetic(n_epochs=300, n_chan=9, n_slots=16, samples_per_epoch=260000, seed=3)
chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 0
with pkg.SynthEngine(samples_per_epoch=260000, n_slots=16, device=0, chunk_samples=chunk,
                     test_hooks=bool(os.environ.get("GAL_LAT_HOOKS"))) as eng:  # (GAL_LAT_HOOKS=1: the GAL_TEST_HOOKS build, for A/B of its knobs)
    st = None
    eng.run_host(p[:1])
    t = time.perf_counter()
    for e in range(300):
        q = p[e:e+1].copy()
        if e > 0: q["flags"][0, :] = 0
        iq, st, stats = eng.run_host(q, st if e > 0 else None)
    dt = time.perf_counter() - t
print("chunk %d: per-epoch run_host: %.3f ms per epoch (0.1 s of signal) -> %.0fx real time, %.1f Msamples/s" % (chunk, dt/300*1e3, 0.1/(dt/300), 300*0.26/dt))

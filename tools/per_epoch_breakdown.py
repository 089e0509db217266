#!/usr/bin/env python3
"""Where does one per-epoch call (INTEGRATION.md option B) spend its time?  plan / execute / finish / copy."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from __graft_entry__ import load_pkg
pkg = load_pkg()
n = 260000
p = pkg.workloads.make_synthetic(n_epochs=300, n_chan=9, n_slots=16, samples_per_epoch=n, seed=3)
for chunk in (0, 160, 208, 416):
    with pkg.SynthEngine(samples_per_epoch=n, n_slots=16, device=0, chunk_samples=chunk) as eng:
        out = torch.empty(n * 2, dtype=torch.int16, device="cuda")
        host = torch.empty(n * 2, dtype=torch.int16, pin_memory=True)
        st = None
        eng.plan(p[:1]); eng.execute(out.data_ptr()); st, _ = eng.finish()
        tp = te = tf = tc = 0.0
        for e in range(1, 201):
            q = p[e:e+1].copy(); q["flags"][0, :] = 0
            t0 = time.perf_counter(); eng.plan(q, st)
            t1 = time.perf_counter(); eng.execute(out.data_ptr())
            t2 = time.perf_counter(); st, stats = eng.finish()
            t3 = time.perf_counter(); host.copy_(out); 
            t4 = time.perf_counter()
            tp += t1 - t0; te += t2 - t1; tf += t3 - t2; tc += t4 - t3
        print("chunk %4d (%d): plan %.3f  execute(host) %.3f  finish(wait) %.3f  d2h %.3f  total %.3f ms; ms_walk %.3f ms_synth %.3f" % (
            chunk, stats["chunk_samples"], tp / 200 * 1e3, te / 200 * 1e3, tf / 200 * 1e3, tc / 200 * 1e3, (tp + te + tf + tc) / 200 * 1e3,
            stats["ms_walk"], stats["ms_synth"]))

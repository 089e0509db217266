#!/usr/bin/env python3
"""k_synth_g's window forms over sample rates: 12 channels, 300 epochs of 0.1 s, one handle, the default path against
GAL_CFG_EXACT_REPLAY (k_synth).   python tools/rate_sweep.py [rates in MS/s ...]"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from __graft_entry__ import load_pkg
pkg = load_pkg()
rates = [float(a) * 1e6 for a in sys.argv[1:]] or [2.6e6, 3.0e6, 4.0e6, 5.0e6, 6.5e6, 8e6, 16e6, 25e6]
E = 300
for rate in rates:
    n = int(round(rate / 10))
    p = pkg.workloads.make_synthetic(n_epochs=E, n_chan=12, n_slots=16, samples_per_epoch=n, sample_rate=rate, seed=11)
    row = []
    for flags in (0, pkg.synth.GAL_CFG_EXACT_REPLAY):
        with pkg.SynthEngine(sample_rate=rate, samples_per_epoch=n, n_slots=16, device=0, flags=flags) as eng:
            out = torch.empty(E * n * 2, dtype=torch.int16, device="cuda:0")
            for rep in range(3):
                eng.plan(p); eng.execute(out.data_ptr()); st, stats = eng.finish()
            torch.cuda.synchronize()
            t = time.perf_counter()
            reps = 10
            for rep in range(reps):
                eng.plan(p); eng.execute(out.data_ptr()); st, stats = eng.finish()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t) / reps
            row.append((dt, stats))
            chk = int(out.view(torch.int32).sum(dtype=torch.int64).item()) & 0xffffffff
            row.append(chk)
    (d0, s0), c0, (d1, s1), c1 = row
    print("%5.2f MS/s: family %d form %d  %7.3f ms (kernel %6.3f) = %6.1f G samples/s | exact replay %7.3f ms (kernel %6.3f) = %6.1f G | x%.2f  same bytes: %s"
          % (rate / 1e6, s0["kernel_family"], s0["window_mode"], d0 * 1e3, s0["ms_synth"], E * n / d0 / 1e9, d1 * 1e3, s1["ms_synth"], E * n / d1 / 1e9, d1 / d0, c0 == c1))

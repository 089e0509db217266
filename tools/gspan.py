#!/usr/bin/env python3
"""What the carrier gathers' LDS bank conflicts cost k_synth_g: the kernel's time on M-SYN12's geometry with the Doppler range
narrowed.  At |f| <= 300 Hz the 32 lanes of an LDS pass read at most 32 CONSECUTIVE table entries (16 x 511 |d| < 1 per lane):
no conflicts; the instruction stream is the same.  python tools/gspan.py  (GPU; DESIGN.md section 5.1, 'The carrier gathers and the LDS')"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_pkg
import torch
pkg = load_pkg()
for span in (3500.0, 1500.0, 600.0, 300.0, 100.0):
    p = pkg.workloads.make_synthetic(n_epochs=1199, n_chan=12, n_slots=16, seed=20241008, doppler_span=span)
    with pkg.SynthEngine(device=0) as eng:
        eng.plan(p)
        buf = torch.empty(eng.output_bytes() // 2, dtype=torch.int16, device="cuda")
        for i in range(3):
            eng.execute(buf.data_ptr()); st, stats = eng.finish()
        ts = []
        for i in range(10):
            eng.execute(buf.data_ptr()); st, stats = eng.finish(); ts.append(stats["ms_synth"])
    print("doppler span +-%5.0f Hz: family %d, repaired %6d, k_synth_g alone %.4f ms (min %.4f), walkers %.3f ms" % (
        span, stats["kernel_family"], stats["repaired_groups"], float(np.mean(ts)), float(np.min(ts)), stats["ms_walk"]), flush=True)

#!/usr/bin/env python3
"""Cost of the first batch after gal_synth_create (code-object loading happens in create): python tools/first_call_probe.py"""
import sys, time, os
sys.path.insert(0, "."); 
import numpy as np, torch
from __graft_entry__ import load_pkg
pkg = load_pkg()
p = pkg.workloads.make_synthetic(n_epochs=8, n_chan=9, n_slots=16, samples_per_epoch=260000, seed=3)
t0 = time.perf_counter()
eng = pkg.SynthEngine(samples_per_epoch=260000, n_slots=16, device=0)
t1 = time.perf_counter()
eng.run_host(p); t2 = time.perf_counter()
eng.run_host(p); t3 = time.perf_counter()
print("create %.1f ms, first run_host %.1f ms, second %.1f ms" % ((t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3))

#!/usr/bin/env python3
"""Durations of the walker kernels for a full execute and for execute_range(E-1, 1) (everything in front of the last epoch
walked silently: no checkpoint stores) -- run under rocprofv3 --kernel-trace by tools/walk_silent_probe.sh."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from __graft_entry__ import load_pkg
pkg = load_pkg()
p = pkg.workloads.m_syn12()
E = p.shape[0]
with pkg.SynthEngine(samples_per_epoch=260000, n_slots=16, device=0) as eng:
    eng.plan(p)
    out = torch.empty(eng.output_bytes() // 2, dtype=torch.int16, device="cuda")
    for _ in range(3):
        eng.execute(out.data_ptr()); eng.finish()
    torch.cuda.synchronize()
    for _ in range(3):
        eng.execute(out.data_ptr(), E - 1, 1); eng.finish()

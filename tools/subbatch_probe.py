#!/usr/bin/env python3
"""Timing probe (MI355X): what would one handle gain if execute() cut its batch into K sub-batches whose walker chains
run beside the synthesis of the sub-batch before?  Emulated with K independent handles over K slices of M-SYN12 (every
slice restarts its channels, so the IQ of slices > 0 is NOT the scenario's -- timing only)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from __graft_entry__ import load_pkg
pkg = load_pkg()
p = pkg.workloads.m_syn12()
E = p.shape[0]
n = 260000
for K in (1, 2, 3, 4, 6):
    bounds = np.linspace(0, E, K + 1).astype(int)
    engs, outs = [], []
    for a, b in zip(bounds[:-1], bounds[1:]):
        q = p[a:b].copy()
        q["flags"][0, :12] = 1
        q["carr_phase0"][0, :12] = 0.25
        q["page_init"][0, :12] = p["page_init"][0, :12]
        e = pkg.SynthEngine(samples_per_epoch=n, n_slots=16, device=0)
        e.plan(q)
        engs.append(e)
        outs.append(torch.empty(e.output_bytes() // 2, dtype=torch.int16, device="cuda"))
    def step():
        for e, o in zip(engs, outs):
            e.execute(o.data_ptr())
        for e in engs:
            e.finish()
    for _ in range(3): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    N = 30
    for _ in range(N): step()
    torch.cuda.synchronize()
    print("K=%d sub-batches: %.3f ms per full batch" % (K, (time.perf_counter() - t0) / N * 1e3))
    for e in engs: e.close()

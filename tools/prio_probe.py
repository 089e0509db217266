#!/usr/bin/env python3
"""Probe (MI355X): can a LOW-priority third handle fill the tail of the normal-priority synthesis kernels?
1199 one-epoch blocks on 768 block slots leave the second round of every k_synth launch 56 % full; two handles of
equal priority only stretch each other.  Here handles 0,1 (normal priority, ping-pong) are complemented by handle 2
on a low-priority stream; completion is out of order, the host polls."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from __graft_entry__ import load_pkg
pkg = load_pkg()
p = pkg.workloads.m_syn12()
n = 260000
print("priority range", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "?")

def make(prio):
    e = pkg.SynthEngine(samples_per_epoch=n, n_slots=16, device=0)
    st = torch.cuda.Stream(priority=prio)
    e.set_stream(st.cuda_stream)
    e.plan(p)
    o = torch.empty(e.output_bytes() // 2, dtype=torch.int16, device="cuda")
    return e, st, o

def run(prios, total):
    hs = [make(pr) for pr in prios]
    evs = [None] * len(hs)
    done = 0
    submitted = 0
    t0 = None
    def submit(i):
        e, st, o = hs[i]
        e.execute(o.data_ptr())
        ev = torch.cuda.Event()
        ev.record(st)
        evs[i] = ev
    # warm-up
    for i in range(len(hs)):
        submit(i)
    for i in range(len(hs)):
        hs[i][0].finish(); evs[i] = None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    counts = [0] * len(hs)
    for i in range(len(hs)):
        submit(i); submitted += 1
    while done < total:
        for i in range(len(hs)):
            if evs[i] is not None and evs[i].query():
                hs[i][0].finish(); evs[i] = None; done += 1; counts[i] += 1
                if submitted < total:
                    submit(i); submitted += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("prios %s: %.4f ms per step (%d steps, per handle %s)" % (prios, dt / total * 1e3, total, counts))
    for e, st, o in hs: e.close()

run([0, 0], 100)
run([0, 0, 1], 150)
run([0, 0, 0], 150)
run([0, 1], 100)
run([-1, 0], 100)
run([0, 0, 1, 1], 200)

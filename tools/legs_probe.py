import os, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from __graft_entry__ import load_pkg
pkg = load_pkg()
p = pkg.workloads.m_syn12()
n = 260000
for legs in (4, 8, 12, 16, 24):
    os.environ["GAL_WALK_LEGS"] = str(legs)
    with pkg.SynthEngine(samples_per_epoch=n, n_slots=16, device=0, test_hooks=True) as eng:
        eng.plan(p)
        out = torch.empty(eng.output_bytes() // 2, dtype=torch.int16, device="cuda")
        ws = []
        for _ in range(8):
            eng.execute(out.data_ptr()); st, stats = eng.finish(); ws.append(stats["ms_walk"])
        t0 = time.perf_counter()
        for _ in range(20):
            eng.execute(out.data_ptr()); eng.finish()
        dt = (time.perf_counter() - t0) / 20 * 1e3
        print("legs %2d: ms_walk %.3f  step %.3f ms  passes %d  counts %s" % (legs, sorted(ws)[len(ws)//2], dt, stats["walk_passes"], eng.walk_counts()))

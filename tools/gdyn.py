import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from __graft_entry__ import load_pkg
pkg = load_pkg()
for name, p in (("dyn", pkg.workloads.m_dyn(2999)), ("syn12", pkg.workloads.m_syn12(1199))):
    for flags in (0, 4):
        with pkg.SynthEngine(device=0, flags=flags) as eng:
            eng.plan(p)
            buf = torch.empty(eng.output_bytes() // 2, dtype=torch.int16, device="cuda")
            for i in range(4):
                t0 = time.perf_counter(); eng.execute(buf.data_ptr()); st, stats = eng.finish(); dt = time.perf_counter() - t0
            print(name, "flags", flags, {k: stats[k] for k in ("kernel_family", "repaired_groups", "ms_synth", "ms_walk", "ms_repair", "walk_passes", "synth_runs")}, "wall %.3f ms" % (dt * 1e3), flush=True)

#!/usr/bin/env python3
"""How many walker passes fresh scenarios need: M-SYN12-sized batches of 32 seeds on ONE handle, planned
asynchronously; per seed: walk passes, synthesis runs, legs walked / translated, plan and upload time."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from __graft_entry__ import load_pkg  # noqa: E402

pkg = load_pkg()
n, E = 260000, int(sys.argv[1]) if len(sys.argv) > 1 else 1199
dyn = len(sys.argv) > 2 and sys.argv[2] == "dyn"
out = torch.empty(E * n * 2, dtype=torch.int16, device="cuda")
with pkg.SynthEngine(samples_per_epoch=n, n_slots=16, device=0) as eng:
    for k in range(32):
        p = pkg.workloads.make_synthetic(n_epochs=E, n_chan=12, n_slots=16, samples_per_epoch=n, seed=1000 + k, dyn_track=dyn)
        t0 = time.perf_counter()
        eng.plan(p, wait=False)
        eng.execute(out.data_ptr())
        st, stats = eng.finish()
        dt = (time.perf_counter() - t0) * 1e3
        w, tr, fb = eng.walk_counts()
        print("seed %d: passes %d synth_runs %d walked %d translated %d fallbacks %d  plan %.3f h2d %.3f walk %.3f synth %.3f total %.3f ms" % (
            1000 + k, stats["walk_passes"], stats["synth_runs"], w, tr, fb, stats["ms_plan"], stats["ms_h2d"], stats["ms_walk"],
            stats["ms_synth"], dt))

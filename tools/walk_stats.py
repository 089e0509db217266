#!/usr/bin/env python3
"""Carrier-walker statistics on the headline workload: passes, legs walked / translated, walker ms."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401  (initialise HIP through torch first)
from __graft_entry__ import load_pkg

pkg = load_pkg()
for name, kw in [("syn12", {}), ("dyn", {"dyn_track": True})]:
    params = pkg.shard.rank_workload(0, 1199, n_chan=12, n_slots=16, samples_per_epoch=260000, sample_rate=2.6e6, **kw)
    eng = pkg.SynthEngine(sample_rate=2.6e6, samples_per_epoch=260000, n_slots=16, device=0)
    eng.plan(params)
    out = torch.empty(eng.output_bytes() // 2, dtype=torch.int16, device="cuda")
    for it in range(3):
        eng.execute(out.data_ptr())
        st = eng.finish()[1]
    print(name, "passes", st["walk_passes"], "walked/translated/fallbacks", eng.walk_counts(), "legs", 1199 * 8 * 12,
          "ms_walk %.3f ms_synth %.3f" % (st["ms_walk"], st["ms_synth"]))

import sys, os, time
sys.path.insert(0, ".")
import numpy as np, torch
from __graft_entry__ import load_pkg
pkg = load_pkg()
rate = float(sys.argv[1]); n = int(rate / 10); E = int(sys.argv[2])
p = pkg.workloads.make_synthetic(n_epochs=E, n_chan=12, n_slots=16, samples_per_epoch=n, sample_rate=rate, seed=7)
for env in ("1", "0"):
    os.environ["GAL_SYNTH_RW"] = env
    with pkg.SynthEngine(sample_rate=rate, samples_per_epoch=n, n_slots=16, device=0, test_hooks=True) as eng:
        eng.plan(p)
        out = torch.empty(eng.output_bytes() // 2, dtype=torch.int16, device="cuda")
        for _ in range(3):
            eng.execute(out.data_ptr()); st, stats = eng.finish()
        ms = []
        for _ in range(10):
            eng.execute(out.data_ptr()); st, stats = eng.finish(); ms.append(stats["ms_synth"])
        print("rate %.4g MS/s mode %d: k_synth %.3f ms (%.1f G samples/s)" % (rate / 1e6, stats["window_mode"], sorted(ms)[5], E * n / sorted(ms)[5] / 1e6))

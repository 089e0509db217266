#!/usr/bin/env python3
"""Quick GPU check of the group kernel (k_synth_g): parity against the oracle on a few epochs of M-SYN12 and of a 9-SV
batch, then the kernel time of the full 1199-epoch batch for both families."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_pkg
from oracle_binding import oracle_run
import torch
pkg = load_pkg()
ok = True
for nch, nep, seed in ((12, 4, 1), (9, 3, 2), (5, 2, 3), (1, 2, 4), (12, 3, 5)):
    p = pkg.workloads.make_synthetic(n_epochs=nep, n_chan=nch, n_slots=16, seed=seed)
    ref, rst = oracle_run(p, 260000, 2.6e6)
    with pkg.SynthEngine(device=0) as eng:
        iq, st, stats = eng.run_host(p)
    bad = int(np.count_nonzero(iq != ref))
    print(nch, nep, "family", stats["kernel_family"], "repaired", stats["repaired_groups"], "mismatch", stats["chain_mismatch"], "bad", bad,
          "state", np.array_equal(st["carr_phase"].view(np.uint64), rst["carr_phase"].view(np.uint64)), flush=True)
    if bad:
        idx = np.flatnonzero(iq != ref)[:10]
        print("  first bad int16 indices", idx, "samples", idx // 2, "groups", (idx // 2) // 16)
        ok = False
if "--time" in sys.argv:
    p = pkg.workloads.m_syn12(1199)
    for flags in (0, 4):
        with pkg.SynthEngine(device=0, flags=flags) as eng:
            eng.plan(p)
            buf = torch.empty(eng.output_bytes() // 2, dtype=torch.int16, device="cuda")
            for i in range(3):
                eng.execute(buf.data_ptr()); st, stats = eng.finish()
            ts = []
            for i in range(10):
                eng.execute(buf.data_ptr()); st, stats = eng.finish(); ts.append(stats["ms_synth"])
            print("flags", flags, "family", stats["kernel_family"], "repaired", stats["repaired_groups"], "ms_synth", np.round(ts, 4), "ms_walk", stats["ms_walk"],
                  "checksum", hex(int(buf.view(torch.int32).sum().item()) & 0xffffffff), flush=True)
sys.exit(0 if ok else 1)

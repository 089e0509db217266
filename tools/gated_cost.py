#!/usr/bin/env python3
"""What one record the group kernel cannot take costs a batch: M-SYN12 as it is, and with ONE of its twelve
channels creeping (f_carr = 1e-9 Hz in the first 300 epochs: those 300 records go to an accumulating exact-replay launch behind k_synth_g, the
other eleven channels stay on it) or standing still (f_carr = 0: stays on k_synth_g), one handle, 20 steps each, alternating.  GPU.  python tools/gated_cost.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from __graft_entry__ import load_pkg  # noqa: E402

pkg = load_pkg()
p = pkg.workloads.m_syn12()
q = p.copy()
q["f_carr"][:300, 5] = 1e-9  # (in a quarter of the epochs: in more than half of them the whole batch takes the exact-replay kernel)
q["f_code"][:300, 5] = 1.023e6
z = p.copy()
z["f_carr"][:, 5] = 0.0
z["f_code"][:, 5] = 1.023e6
n = 260000


def run(params, steps=20):
    with pkg.SynthEngine(samples_per_epoch=n, n_slots=params.shape[1], device=0) as eng:
        eng.plan(params)
        out = torch.empty(eng.output_bytes() // 2, dtype=torch.int16, device="cuda")
        for _ in range(3):
            eng.execute(out.data_ptr())
            eng.finish()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.execute(out.data_ptr())
            st, stats = eng.finish()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3, stats


for rep in range(3):
    a, sa = run(p)
    b, sb = run(q)
    c, sc = run(z)
    print("  (still: synth %.3f ms, repair %.3f ms, %d groups replayed; creeping: synth %.3f, repair %.3f, %d groups)" % (
        sc["ms_synth"], sc["ms_repair"], sc["repaired_groups"], sb["ms_synth"], sb["ms_repair"], sb["repaired_groups"]))
    print("ungated %.4f ms (family %d, exact records %d)   one creeping channel %.4f ms (family %d, exact records %d) ratio %.3f   "
          "one still channel %.4f ms (family %d, exact records %d) ratio %.3f" % (
              a, sa["kernel_family"], sa["exact_records"], b, sb["kernel_family"], sb["exact_records"], b / a,
              c, sc["kernel_family"], sc["exact_records"], c / a))

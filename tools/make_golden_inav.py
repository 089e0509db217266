#!/usr/bin/env python3
"""Build tests/golden/inav_tv_20feb2022.csv from the recorded I/NAV pages the reference holds.

The only golden data harshadms/galileo-sdr-sim itself carries for this path is tv/<date>/<svid>.csv: one row
per 2 s page, `TOW,WN,SVID,<60 hex digits>` = the 240 page bits before channel coding (even half 114 + 6
tail, odd half 114 + 6 tail) as broadcast on 20 Feb 2022 from GST 08:00:01.  rinex_files/20feb2022.rnx (our
tests/golden/20feb2022.rnx) carries the ephemerides of the same day, so the reference's page generator
(src/inav-msg.cpp:170-411), fed from that RINEX file, must reproduce every field it derives from the file.

This script only SELECTS rows (data, not code) and copies them verbatim; it runs in the build container,
where /root/reference exists.  tests/test_inav_kat.py consumes the fixture.

    python tools/make_golden_inav.py [--ref /root/reference] [--rows 150]
"""
import argparse
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TV_DIR = "tv/20_FEB_2022_GST_08_00_01"
# SVs whose tv rows carry IODnav values present in 20feb2022.rnx for that hour (others started the hour on
# a batch older than the file's first record)
SVIDS = (1, 4, 9, 13, 24, 31)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--rows", type=int, default=150, help="rows (2 s pages) per SV: 150 = ten 30 s subframes")
    args = ap.parse_args()
    out_path = os.path.join(ROOT, "tests", "golden", "inav_tv_20feb2022.csv")
    n = 0
    with open(out_path, "w") as out:
        out.write("# source: %s/<svid>.csv of harshadms/galileo-sdr-sim (recorded broadcast I/NAV pages), first %d rows\n"
                  % (TV_DIR, args.rows))
        out.write("# columns: TOW,WN,SVID,240 page bits as 60 hex digits (even half 114+6 tail | odd half 114+6 tail)\n")
        for sv in SVIDS:
            with open(os.path.join(args.ref, TV_DIR, "%d.csv" % sv)) as f:
                for i, line in enumerate(f):
                    if i >= args.rows:
                        break
                    tow, wn, svid, hx = line.strip().split(",")
                    assert int(svid) == sv and len(hx) == 60
                    out.write("%s,%s,%s,%s\n" % (tow, wn, svid, hx))
                    n += 1
    print("wrote %s: %d rows" % (out_path, n))


if __name__ == "__main__":
    main()

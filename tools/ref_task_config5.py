#!/usr/bin/env python3
"""BASELINE config 5's per-rank units through THE REFERENCE PROGRAM (oracle/_ref/ref_task, the reference's file-sink program compiled
here from its own text): the eight static sites of galileo-sdr-sim_amd/shard.py LOCATIONS, 300 s each from 2022/02/20,12:00:00,
iono as the reference runs it.  Records the md5 and the byte count of every file in tests/golden/ref_task_config5.json -- the answers
tests/test_cli.py::test_cli_config5_all_eight_sites holds the product's `--sites` run to on the GPU box, where /root/reference does
not exist.  CPU only, needs /root/reference (for the build of ref_task); ~3 min per site, --jobs at a time (3.1 GB of file each).

    python tools/ref_task_config5.py [--jobs 4]
"""
import json
import os
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from ref_task_goldens import run_ref_task  # noqa: E402

BIN = os.path.join(ROOT, "oracle", "_ref", "ref_task")
START, DUR = "2022/02/20,12:00:00", 300


def main():
    from __graft_entry__ import load_pkg
    sites = load_pkg().shard.LOCATIONS
    jobs = int(sys.argv[sys.argv.index("--jobs") + 1]) if "--jobs" in sys.argv else 4

    def one(k):
        llh = sites[k]
        args = "-l %.10g,%.10g,%.10g -t %s -d %d" % (llh[0], llh[1], llh[2], START, DUR)
        with tempfile.TemporaryDirectory(dir="/tmp") as d:
            md5, n, dt, rc = run_ref_task(BIN, args, os.path.join(d, "r.bin"), port=21000 + k, timeout=1800)
        print("site %d %s: md5 %s, %d bytes, %.0f s, exit %s" % (k, args, md5, n, dt, rc), flush=True)
        return dict(site=k, llh=list(llh), args=args, md5=md5, bytes=n, exit=rc)

    with ThreadPoolExecutor(jobs) as ex:
        out = list(ex.map(one, range(len(sites))))
    path = os.path.join(ROOT, "tests", "golden", "ref_task_config5.json")
    json.dump(dict(start=START, duration_s=DUR, what="oracle/_ref/ref_task (the reference's file-sink program, reference flags) on the "
                   "eight sites of shard.LOCATIONS: md5 and size of its ishort file", sites=out), open(path, "w"), indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()

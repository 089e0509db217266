#!/usr/bin/env python3
"""Randomised END-TO-END parity against THE REFERENCE PROGRAM ITSELF (oracle/_ref/ref_task: the reference's file-sink program
compiled here from its own text, oracle/ref_task_harness.cpp): random receiver sites, start times over the day of the navigation
file, durations 2-45 s (crossing the 30 s re-allocation), iono on/off, -T on/off.  Per case: the reference writes its ishort
file; the repository's host front-end (libgalscen.so) -> the oracle's loop produces the same scenario on CPU; the two md5s must
be equal.  This checks the front-end (RINEX reader, orbits, ranges, iono, channel allocation, I/NAV pages) and the oracle against
the reference directly -- no recorded md5 in between.  CPU only; needs /root/reference built into oracle/_ref (make -C oracle ref).

    python tools/ref_task_fuzz.py [n_cases] [seed] [jobs] [--cli]

--cli (on the GPU box; the binary travels there with oracle/_ref/): the other side is the PRODUCT -- the galileo-sdr-sim CLI, front-end ->
HIP -> file, on the same command line -- instead of front-end -> oracle; output of the round's run: profiles/archive/r04_ref_task_fuzz_cli.log.
--odd: the corners of the command line too -- -t or -l left out, fractional durations and seconds, the ends of the coordinate ranges,
starts at the edges of the file's span (profiles/archive/r04_ref_task_fuzz_odd.log).
--record F: besides comparing, write every case with the reference's md5 / byte count to the JSON file F (made HERE, where the reference
runs in parallel); --replay F --cli: on the GPU box, take the reference's answers from F instead of running the reference there (it binds
a fixed UDP port, so it runs one instance at a time without network namespaces): only the product runs (r04_ref_task_replay_cli.log).
--toc: long cases (65-130 s) that run across a 10-minute mark of the records' TOC grid and several 30 s refreshes (r04_ref_task_fuzz_toc.log).
--long300: 200-300 s cases, up to the reference's cap of 3000 epochs: ten re-allocations per run (tests/golden/ref_task_recorded_300s.json).

A case our front-end REJECTS (start outside the file's span) is counted as skipped and what the reference did with it is printed (it
exits with status 1 there too).  A case in which a satellite in view runs out of ephemeris is the reference's undefined behaviour
(it indexes its vector with -1; seen: it never finishes) and is counted apart.  Output of the round's run: profiles/archive/r04_ref_task_fuzz.log.  Test infrastructure.
"""
import hashlib
import multiprocessing as mp
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
NAV = os.path.join(ROOT, "tests", "golden", "20feb2022.rnx")
BIN = os.path.join(ROOT, "oracle", "_ref", "ref_task")


def make_case(rng, c, odd=False):
    lat, lon, h = rng.uniform(-89, 89), rng.uniform(-180, 180), rng.uniform(0, 4000)
    if rng.random() < 0.15:  # round numbers: the defaults people type
        lat, lon, h = float(round(lat)), float(round(lon)), 100.0
    hh, mm, ss = int(rng.integers(0, 24)), int(rng.integers(0, 60)), int(rng.integers(0, 60))
    dur = float(rng.choice([2, 3, 5, 8, 12, 20, 31, 45], p=[.15, .15, .2, .15, .15, .1, .05, .05]))
    k = dict(c=c, llh=(lat, lon, h), start="2022/02/20,%02d:%02d:%02d" % (hh, mm, ss), dur=dur,
             iono=bool(rng.integers(0, 2)), tovr=bool(rng.random() < 0.15), no_t=False, no_l=False)
    if odd:  # --odd: the corners of the command line -- options left out, fractions, the ends of the coordinate ranges
        r = rng.random()
        if r < 0.12:
            k["no_t"], k["tovr"] = True, False  # start = the file's first TOC (src/gnss-time.cpp:159-165)
        elif r < 0.24:
            k["no_l"] = True  # the default site (src/main.cpp:187-189)
        elif r < 0.44:
            k["dur"] = float(round(rng.uniform(1.2, 9.0), int(rng.integers(1, 4))))  # (int)(d * 10 + 0.5) epochs (src/main.cpp:274-275)
        elif r < 0.56:
            k["start"] += ".%d" % rng.integers(1, 10)  # seconds are floored (src/main.cpp:269)
        elif r < 0.70:
            k["llh"] = (float(rng.choice([-90, 90, 0])), float(rng.choice([-180, 180, 0])), float(rng.choice([-400, 0, 9000, 2.0e7])))
        elif r < 0.80:
            k["start"] = "2022/02/19,%02d:%02d:%02d" % (22 + int(rng.integers(0, 2)), mm, ss)  # the file's first two hours
        elif r < 0.90:
            k["start"] = "2022/02/20,23:%02d:%02d" % (int(rng.integers(20, 31)), 0 if rng.random() < 0.5 else ss)  # around tmax
            k["dur"] = float(rng.choice([2, 3, 5]))
    if "--toc" in sys.argv[1:]:  # --toc: every case runs across a 10-minute mark (the records' TOC grid: epoch_matcher moves on at a
        # 30 s refresh behind it, src/galileo-sdr.cpp:545-562) and across two to four refreshes
        k["start"] = "2022/02/20,%02d:%d9:%02d" % (hh, int(rng.integers(0, 6)), int(rng.integers(0, 50)))
        k["dur"] = float(rng.choice([65, 80, 100, 130]))
        k["tovr"] = False
    if "--long300" in sys.argv[1:]:  # --long300 (round 6): 200-300 s, up to the reference's cap (USER_MOTION_SIZE 3000 epochs,
        # include/constants.h:14): six to ten 30 s re-allocations and several TOC marks per run
        k["dur"] = float(rng.choice([200, 240, 270, 300]))
        k["tovr"] = bool(rng.random() < 0.1)
    return k


def case_args(k):
    return " ".join(([] if k["no_l"] else ["-l %.9g,%.9g,%.9g" % k["llh"]]) +
                    ([] if k["no_t"] else ["-%s %s" % ("T" if k["tovr"] else "t", k["start"])]) +
                    ["-d %g" % k["dur"]] + ([] if k["iono"] else ["-I 1"]))


CLI = os.path.join(ROOT, "galileo-sdr-sim_amd", "galileo-sdr-sim")
USE_CLI = "--cli" in sys.argv[1:]


def md5_file(path):
    h = hashlib.md5()
    n = 0
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""):
            h.update(blk)
            n += len(blk)
    return h.hexdigest(), n


def run_case_cli(k, args, ref_md5, ref_n, rc):
    """The product's side of a case: the CLI on the reference's command line (-P 0: no position listener)."""
    with tempfile.TemporaryDirectory(dir="/tmp") as d:
        out = os.path.join(d, "o.bin")
        # (the very same arguments, -T included: plain -T is the reference as built)
        r = subprocess.run([CLI, "-e", NAV] + args.split() + ["-P", "0", "-o", out], capture_output=True, text=True)
        if r.returncode != 0:
            if "Invalid start time" in r.stderr:
                return dict(k=k, args=args, status="skipped", why=r.stderr.strip().splitlines()[-1], ref_n=ref_n, ref_rc=rc)
            return dict(k=k, args=args, status="DIFFERENT", ours="exit %d: %s" % (r.returncode, r.stderr[-200:]), ref=ref_md5, n=0, ref_n=ref_n)
        ours, n = md5_file(out)
    if "no ephemeris within an hour" in r.stderr:
        return dict(k=k, args=args, status="undefined", ref=ref_md5, ref_n=ref_n, gaps=r.stderr.count("no ephemeris within an hour"))
    return dict(k=k, args=args, status="equal" if (ours == ref_md5 and n == ref_n) else "DIFFERENT", ours=ours, ref=ref_md5, n=n, ref_n=ref_n,
                n_sv=-1, samples=n // 4)


def _flag_value(name):
    a = sys.argv[1:]
    return a[a.index(name) + 1] if name in a and a.index(name) + 1 < len(a) else None


def run_case(k):
    from ref_task_goldens import run_ref_task
    if "recorded" in k:  # --replay: the reference's answer comes from the file
        ref_md5, ref_n, rc = k["recorded"]
        return run_case_cli(k, case_args(k), ref_md5, ref_n, rc)
    from __graft_entry__ import load_pkg
    from oracle_binding import oracle_run
    pkg = load_pkg()
    args = case_args(k)
    # the front-end reads -l through the same text (sscanf %lf of what the command line says)
    llh = (42.3601, -71.0589, 2.0) if k["no_l"] else tuple(float(v) for v in args.split()[1].split(","))
    with tempfile.TemporaryDirectory(dir="/tmp") as d:
        out = os.path.join(d, "r.bin")
        ref_md5, ref_n, dt, rc = run_ref_task(BIN, args, out, port=20000 + k["c"] % 20000, timeout=120 + 5 * k["dur"])
    if USE_CLI:
        return run_case_cli(k, args, ref_md5, ref_n, rc)
    try:
        sc = pkg.Scenario(NAV, llh=llh, start=None if k["no_t"] else k["start"], duration_s=k["dur"], iono_enable=k["iono"], time_overwrite="ref" if k["tovr"] else False)
        rows = sc.all()
        gaps = sc.eph_gaps
    except pkg.GalScenError as e:
        return dict(k=k, args=args, status="skipped", why=str(e), ref_n=ref_n, ref_rc=rc)
    if gaps:
        # a satellite in view ran out of ephemeris: the reference reads eph_vector[sv][-1] there (src/galileo-sdr.cpp:458,555-558) --
        # undefined behaviour (seen: xyz2llh never returns on what it read); the front-end's policy is galscen.h strict_eph
        return dict(k=k, args=args, status="undefined", ref=ref_md5, ref_n=ref_n, gaps=gaps)
    iq, _ = oracle_run(rows, 260000, 2.6e6)
    ours = hashlib.md5(iq.tobytes()).hexdigest()
    n_sv = int((rows["prn"] > 0).sum(axis=1).max())
    return dict(k=k, args=args, status="equal" if (ours == ref_md5 and iq.nbytes == ref_n) else "DIFFERENT", ours=ours, ref=ref_md5,
                n=iq.nbytes, ref_n=ref_n, n_sv=n_sv, samples=rows.shape[0] * 260000)


def main():
    rec_path, rep_path = _flag_value("--record"), _flag_value("--replay")
    argv = [a for a in sys.argv[1:] if not a.startswith("--") and a not in (rec_path, rep_path)]
    n_cases = int(argv[0]) if len(argv) > 0 else 40
    seed = int(argv[1]) if len(argv) > 1 else 1
    jobs = int(argv[2]) if len(argv) > 2 else 6
    if rep_path:
        import json
        cases = json.load(open(rep_path))
        for k in cases:
            k["llh"] = tuple(k["llh"])
        n_cases = len(cases)
    else:
        if not os.path.exists(BIN):
            sys.exit("oracle/_ref/ref_task is not built (make -C oracle ref, with /root/reference present)")
        rng = np.random.default_rng(seed)
        cases = [make_case(rng, c, odd="--odd" in sys.argv[1:]) for c in range(n_cases)]
    recorded = []
    t0 = time.time()
    bad = skipped = undefined = 0
    samples = 0
    svs = {}
    with mp.get_context("spawn").Pool(jobs) as pool:
        for r in pool.imap_unordered(run_case, cases):
            if rec_path and r.get("ref") not in ("timeout",) and "ref_n" in r:
                recorded.append(dict(r["k"], recorded=[r.get("ref"), r["ref_n"], r.get("ref_rc", -6)]))
            if r["status"] == "skipped":
                skipped += 1
                print("skipped case %d [%s]: %s; the reference wrote %d bytes (exit %d)" % (r["k"]["c"], r["args"], r["why"], r["ref_n"], r["ref_rc"]), flush=True)
                continue
            if r["status"] == "undefined":
                undefined += 1
                print("undefined case %d [%s]: a satellite in view runs out of ephemeris (%d time(s)): the reference indexes out of bounds "
                      "there; it %s" % (r["k"]["c"], r["args"], r["gaps"], "did not finish (%d bytes written)" % r["ref_n"] if r["ref"] == "timeout"
                                        else "wrote %d bytes" % r["ref_n"]), flush=True)
                continue
            svs[r["n_sv"]] = svs.get(r["n_sv"], 0) + 1
            samples += r["samples"]
            if r["status"] != "equal":
                bad += 1
                print("DIFFERENT case %d [%s]: ours %s (%d B)  reference %s (%d B)" % (r["k"]["c"], r["args"], r["ours"], r["n"], r["ref"], r["ref_n"]), flush=True)
    print("ref_task fuzz%s (seed %d): %d cases, %d compared (%.1f M samples), %d different, %d skipped (both reject the start time), %d where "
          "the reference's behaviour is undefined (ephemeris gap), SV counts %s, %.0f s" % (
              " against the product CLI (front-end -> HIP -> file)" if USE_CLI else " against front-end -> oracle", seed, n_cases, n_cases - skipped - undefined, samples / 1e6, bad, skipped, undefined, dict(sorted(svs.items())), time.time() - t0))
    if rec_path:
        import json
        recorded.sort(key=lambda k: k["c"])
        json.dump(recorded, open(rec_path, "w"), indent=0)
        print("recorded %d cases with the reference's md5 in %s" % (len(recorded), rec_path))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()

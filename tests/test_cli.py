"""The CLI keeps the reference's option surface and ishort file contract (src/main.cpp:216-326,
src/galileo-sdr.cpp:326-341,536-542)."""
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "galileo-sdr-sim_amd", "galileo-sdr-sim")
G = os.path.join(ROOT, "tests", "golden")
NAV = os.path.join(G, "20feb2022.rnx")
REF = json.load(open(os.path.join(G, "reference_md5.json")))


def test_cli_usage_and_errors(pkg):
    assert os.path.exists(CLI)
    r = subprocess.run([CLI], capture_output=True, text=True)
    assert r.returncode == 1 and "Usage" in r.stdout
    r = subprocess.run([CLI, "-o", "/tmp/x.bin", "-d", "1"], capture_output=True, text=True)
    assert r.returncode == 1 and "ephemeris/nav_msg file is not specified" in r.stdout
    r = subprocess.run([CLI, "-e", "/nonexistent.rnx", "-d", "1"], capture_output=True, text=True)
    assert r.returncode == 1
    r = subprocess.run([CLI, "-e", NAV, "-t", "2019/01/01,00:00:00", "-d", "1"], capture_output=True, text=True)
    assert r.returncode == 1 and "Invalid start time" in r.stderr


@pytest.mark.gpu
def test_cli_g1_file_md5_equals_reference_output(tmp_path):
    out = tmp_path / "g1.ishort"
    # the reference command line, unchanged (README.md:63 style): -U 1 -b 1 are accepted
    r = subprocess.run([CLI, "-e", NAV, "-l", "-6,51,100", "-t", "2022/02/20,12:00:00", "-d", "10", "-U", "1", "-b",
                        "1", "-I", "1", "-o", str(out), "-B", "32"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    data = out.read_bytes()
    assert len(data) == REF["G1"]["bytes"]
    assert hashlib.md5(data).hexdigest() == REF["G1"]["md5"]


@pytest.mark.gpu
def test_cli_stdout_sink():
    r = subprocess.run([CLI, "-e", NAV, "-l", "-6,51,100", "-t", "2022/02/20,12:00:00", "-d", "10", "-I", "1", "-o",
                        "-"], capture_output=True)
    assert r.returncode == 0
    assert hashlib.md5(r.stdout).hexdigest() == REF["G1"]["md5"]


@pytest.mark.gpu
def test_cli_realtime_pacing_and_live_position(tmp_path):
    """-r paces the output to real time (the reference's FIFO hand-over, src/fifo.cpp, src/galileo-sdr.cpp:570-595);
    a position datagram on the -P port moves the receiver for the epochs that follow (include/socket.h:165-180).
    Which epoch sees the update depends on wall-clock time, so only the frame of the result is checked exactly: the
    epochs before the update are the static scenario's, the tail is not, the length is unchanged."""
    import socket
    import struct
    import threading
    import time

    import numpy as np

    s = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    static = tmp_path / "static.ishort"
    r = subprocess.run([CLI, "-e", NAV, "-l", "-6,51,100", "-t", "2022/02/20,12:00:00", "-d", "3", "-P", "0", "-o", str(static)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr

    def send_later():
        time.sleep(1.5)
        tx = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
        tx.sendto(struct.pack("<3d", -6.01, 51.01, 400.0), ("127.0.0.1", port))

    live = tmp_path / "live.ishort"
    th = threading.Thread(target=send_later)
    t0 = time.perf_counter()
    th.start()
    r = subprocess.run([CLI, "-e", NAV, "-l", "-6,51,100", "-t", "2022/02/20,12:00:00", "-d", "3", "-r", "-P", str(port), "-o",
                        str(live)], capture_output=True, text=True)
    wall = time.perf_counter() - t0
    th.join()
    assert r.returncode == 0, r.stderr
    assert wall >= 2.7  # 29 epochs of 0.1 s, paced
    a = np.fromfile(str(static), dtype=np.int16).reshape(29, -1)
    b = np.fromfile(str(live), dtype=np.int16).reshape(29, -1)
    same = "".join("=" if np.array_equal(a[e], b[e]) else "x" for e in range(29))
    assert "Location Update" in r.stderr, r.stderr[-400:]
    # (the datagram leaves 1.5 s after the process was started; how much of that is start-up varies with the box)
    assert same.startswith("=====") and same.endswith("xxxxx") and "x=" not in same, same


@pytest.mark.gpu
def test_cli_time_overwrite(pkg, tmp_path):
    """-T <date,time> (src/main.cpp:237-257, src/gnss-time.cpp:105-137) through the whole CLI.  Plain -T is the reference as built
    (galscen.h time_overwrite 1; ADVICE r5: the same command line must give the reference's bytes): -t without the range check, so a
    start two and a half years after the navigation file -- an error with -t -- runs with an empty sky and a warning on stderr;
    --ref-T spells that default out.  -T --shift-toe is what the option sets out to do (time_overwrite 2): TOC / TOE shifted, a
    valid run whose bytes are the oracle's on the front-end's rows.  --ref-T / --shift-toe without -T are errors."""
    import numpy as np

    from oracle_binding import oracle_run

    when = "2024/10/08,09:30:00"
    r = subprocess.run([CLI, "-e", NAV, "-l", "-6,51,100", "-t", when, "-d", "2", "-P", "0", "-o", str(tmp_path / "x.ishort")],
                       capture_output=True, text=True)
    assert r.returncode == 1 and "Invalid start time" in r.stderr
    out = tmp_path / "t.ishort"
    r = subprocess.run([CLI, "-e", NAV, "-l", "-6,51,100", "-T", when, "--shift-toe", "-d", "2", "-P", "0", "-o", str(out)],
                       capture_output=True, text=True)
    assert r.returncode == 0 and "WARNING: no satellite" not in r.stderr, r.stderr
    rows = pkg.Scenario(NAV, llh=(-6, 51, 100), start=when, duration_s=2, time_overwrite="shift").all()
    assert (rows["prn"] > 0).any()
    ref_iq, _ = oracle_run(rows, 260000, 2.6e6)
    assert np.array_equal(np.fromfile(str(out), dtype=np.int16), ref_iq)
    for extra in ([], ["--ref-T"]):  # plain -T = the reference as built; --ref-T: its explicit spelling
        r = subprocess.run([CLI, "-e", NAV, "-l", "-6,51,100", "-T", when] + extra + ["-d", "2", "-P", "0", "-o", str(out)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert "WARNING: no satellite in view" in r.stderr and "--shift-toe" in r.stderr
        assert not np.fromfile(str(out), dtype=np.int16).any()  # as the reference: nothing in view, 19 epochs of zeros
    # inside the file's span plain -T is -t: the same bytes
    inside = "2022/02/20,12:00:00"
    r = subprocess.run([CLI, "-e", NAV, "-l", "-6,51,100", "-T", inside, "-d", "2", "-P", "0", "-o", str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    rows = pkg.Scenario(NAV, llh=(-6, 51, 100), start=inside, duration_s=2, time_overwrite=True).all()
    assert np.array_equal(rows, pkg.Scenario(NAV, llh=(-6, 51, 100), start=inside, duration_s=2, time_overwrite=1).all())  # True == 1
    ref_iq, _ = oracle_run(rows, 260000, 2.6e6)
    assert np.array_equal(np.fromfile(str(out), dtype=np.int16), ref_iq)
    for flag in ("--ref-T", "--shift-toe"):  # neither means anything without -T
        r = subprocess.run([CLI, "-e", NAV, "-l", "-6,51,100", "-t", "2022/02/20,12:00:00", flag, "-d", "2", "-P", "0", "-o", str(out)],
                           capture_output=True, text=True)
        assert r.returncode == 1 and "needs -T" in r.stdout
    # -T now: the current time is always acceptable
    r = subprocess.run([CLI, "-e", NAV, "-l", "-6,51,100", "-T", "now", "-d", "1", "-P", "0", "-o", "/dev/null"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


@pytest.mark.gpu
def test_cli_sigint_finishes_the_current_epoch(tmp_path):
    """SIGINT: the reference's handler lets the generator finish the epoch in flight and closes the file
    (src/main.cpp sigint_handler); here: exit status 0 and a file that ends on an epoch boundary, shorter than asked."""
    import signal
    import time

    out = tmp_path / "int.ishort"
    p = subprocess.Popen([CLI, "-e", NAV, "-l", "-6,51,100", "-t", "2022/02/20,12:00:00", "-d", "8", "-r", "-P", "0", "-o",
                          str(out)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    t0 = time.time()  # (how long HIP takes to start varies with the box: wait for five paced epochs in the file, not for a fixed time)
    while time.time() - t0 < 20.0 and not (os.path.exists(str(out)) and os.path.getsize(str(out)) >= 5 * 260000 * 4):
        time.sleep(0.05)
    p.send_signal(signal.SIGINT)
    p.wait(timeout=30)
    assert p.returncode == 0, p.stderr.read()
    size = os.path.getsize(str(out))
    assert size % (260000 * 4) == 0 and 5 * 260000 * 4 <= size < 79 * 260000 * 4


@pytest.mark.gpu
def test_cli_sigint_gives_back_the_preallocated_pages():
    """Un-paced run into a regular file: the sink allocates the file's pages ahead of the writes (fallocate, size kept)
    while the device starts up.  SIGINT in mid-run: exit status 0, the file ends on an epoch boundary short of the full
    length, and nothing stays allocated beyond what was written."""
    import signal
    import time

    if not os.path.isdir("/dev/shm"):
        pytest.skip("needs a tmpfs")
    out = "/dev/shm/galsim_sigint_%d.ishort" % os.getpid()
    epoch = 260000 * 4
    try:
        p = subprocess.Popen([CLI, "-e", NAV, "-l", "-6,51,100", "-t", "2022/02/20,12:00:00", "-d", "300", "-P", "0", "-o", out],
                             stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        t0 = time.time()
        while time.time() - t0 < 60 and not (os.path.exists(out) and os.path.getsize(out) > 0) and p.poll() is None:
            time.sleep(0.005)
        p.send_signal(signal.SIGINT)
        p.wait(timeout=60)
        assert p.returncode == 0, p.stderr.read()
        st = os.stat(out)
        assert st.st_size % epoch == 0 and 0 < st.st_size
        if st.st_size == 2999 * epoch:
            pytest.skip("the run was over before the signal arrived")
        assert st.st_blocks * 512 <= st.st_size + (8 << 20), (st.st_blocks * 512, st.st_size)
    finally:
        if os.path.exists(out):
            os.unlink(out)


def test_cli_sites_errors(tmp_path):
    """--sites: a missing or empty site list is an error before anything is started."""
    r = subprocess.run([CLI, "-e", NAV, "--sites", str(tmp_path / "missing.txt"), "-d", "2"], capture_output=True, text=True)
    assert r.returncode == 1 and "cannot read the site list" in r.stderr
    empty = tmp_path / "empty.txt"
    empty.write_text("# nothing here\n\n")
    r = subprocess.run([CLI, "-e", NAV, "--sites", str(empty), "-d", "2"], capture_output=True, text=True)
    assert r.returncode == 1 and "no site" in r.stderr


@pytest.mark.gpu
def test_cli_sites_one_process_per_site(pkg, tmp_path):
    """BASELINE config 5 through the product entry point: `--sites` starts one process per receiver site (here three sites,
    two processes at a time on GPU 0), every site gets its own ishort file, the parent reports the aggregate; each file
    equals the oracle on the front-end's rows for that site."""
    import numpy as np

    from oracle_binding import oracle_run

    sites = [(-6.0, 51.0, 100.0), (45.0, 10.0, 100.0), (0.0, 0.0, 100.0)]
    named = tmp_path / "third.bin"
    lst = tmp_path / "sites.txt"
    lst.write_text("# lat,lon,hgt[,outfile]\n-6,51,100\n45, 10, 100\n0,0,100,%s\n" % named)
    stem = tmp_path / "run.ishort"
    r = subprocess.run([CLI, "-e", NAV, "--sites", str(lst), "-t", "2022/02/20,12:00:00", "-d", "6", "-I", "1", "-o", str(stem),
                        "--gpus", "1", "--per-gpu", "2"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    assert "Sites = 3  failed = 0" in r.stderr
    outs = [tmp_path / "run.site0.ishort", tmp_path / "run.site1.ishort", named]
    for llh, out in zip(sites, outs):
        rows = pkg.Scenario(NAV, llh=llh, start="2022/02/20,12:00:00", duration_s=6, iono_enable=False).all()
        ref_iq, _ = oracle_run(rows, 260000, 2.6e6)
        got = np.fromfile(str(out), dtype=np.int16)
        assert got.size == ref_iq.size == 59 * 520000 and np.array_equal(got, ref_iq), llh
    # a failing site fails the run (start time outside the file) but the report still comes
    r = subprocess.run([CLI, "-e", NAV, "--sites", str(lst), "-t", "2019/01/01,00:00:00", "-d", "2", "-o", str(stem), "--gpus", "1"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 1 and "failed = 3" in r.stderr


def _oracle_md5(pkg, rows, piece=200):
    """md5 of the oracle's ishort stream on these rows, produced in pieces with the channel state carried (host memory: one piece)."""
    from oracle_binding import oracle_run

    h, st, n = hashlib.md5(), None, 0
    for a in range(0, rows.shape[0], piece):
        iq, st = oracle_run(rows[a:a + piece], 260000, 2.6e6, state_in=st)
        h.update(iq.tobytes())
        n += iq.nbytes
    return h.hexdigest(), n


@pytest.mark.gpu
def test_cli_config5_all_eight_sites(pkg, tmp_path):
    """BASELINE config 5 LITERALLY on the one GPU there is: `--sites` with the eight receiver sites of shard.LOCATIONS, 300 s each
    (2999 epochs, 3 118 960 000 bytes per site), two processes at a time on GPU 0.  Every file's md5 must be (a) the md5 of the
    oracle's stream on the front-end's rows for that site and (b) the md5 THE REFERENCE PROGRAM gave for the same command line
    (tests/golden/ref_task_config5.json, recorded by tools/ref_task_config5.py from oracle/_ref/ref_task where /root/reference
    exists).  Eight files are 25 GB: where the temporary directory has less room the sites run in two launches of four."""
    import shutil
    from concurrent.futures import ThreadPoolExecutor

    rec = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_task_config5.json")))
    sites = list(pkg.shard.LOCATIONS)
    assert len(sites) == 8 and [tuple(s["llh"]) for s in rec["sites"]] == [tuple(x) for x in sites]
    start, dur = rec["start"], rec["duration_s"]

    def front_end_oracle(llh):
        rows = pkg.Scenario(NAV, llh=llh, start=start, duration_s=dur, iono_enable=True).all()
        return _oracle_md5(pkg, rows), int((rows["prn"] > 0).sum(axis=1).max())

    with ThreadPoolExecutor(8) as ex:  # (ctypes releases the GIL inside the oracle: the eight sites beside the CLI runs)
        want = [ex.submit(front_end_oracle, llh) for llh in sites]
        free = shutil.disk_usage(str(tmp_path)).free
        halves = [list(range(8))] if free > 40e9 else [[0, 1, 2, 3], [4, 5, 6, 7]]
        got = {}
        for part in halves:
            lst = tmp_path / "sites.txt"
            lst.write_text("".join("%.10g,%.10g,%.10g\n" % sites[k] for k in part))
            stem = tmp_path / "c5.ishort"
            r = subprocess.run([CLI, "-e", NAV, "--sites", str(lst), "-t", start, "-d", str(dur), "-o", str(stem), "--gpus", "1",
                                "--per-gpu", "2"], capture_output=True, text=True, timeout=1500)
            assert r.returncode == 0, r.stderr[-2000:]
            assert "Sites = %d  failed = 0" % len(part) in r.stderr
            for j, k in enumerate(part):
                f = tmp_path / ("c5.site%d.ishort" % j)
                h, n = hashlib.md5(), 0
                with open(f, "rb") as fh:
                    for blk in iter(lambda: fh.read(1 << 24), b""):
                        h.update(blk)
                        n += len(blk)
                got[k] = (h.hexdigest(), n)
                f.unlink()
        want = [w.result() for w in want]
    for k in range(8):
        (md5, n), n_sv = want[k]
        assert n == 2999 * 260000 * 4 and got[k] == (md5, n), (k, sites[k], n_sv)
        assert (rec["sites"][k]["md5"], rec["sites"][k]["bytes"]) == got[k], (k, sites[k], "reference program")


@pytest.mark.gpu
def test_cli_config3_300s_motion_file(pkg, tmp_path):
    """BASELINE config 3 LITERALLY: a 300 s, 10 Hz circular user-motion file through the command line --
    `-u tests/golden/circle_track_300s.csv -t 2022/02/20,12:00:00 -d 300`, the reference's hook for it: src/galileo-sdr.cpp:443-448 --
    2999 epochs, 3 118 960 000 bytes.  The file's md5 must be the md5 of the oracle's stream on the front-end's rows, every byte.
    THE REFERENCE HAS NO -u (getopt accepts it, no case handles it: src/main.cpp:216; galileo_task always runs its static mode,
    src/galileo-sdr.cpp:221-222): tests/golden/ref_task_config3.json (tools/ref_task_config3.py, made where the reference can be built)
    records as a CHECKED fact that the reference program's file for this very command line is the static default-site scenario, byte
    for byte -- so there are no reference bytes for a moving receiver; what the reference does have, the per-epoch position hook
    (xyz[iumd], :443-448) fed over UDP, is what test_live_position.py and test_cli_udp_position_updates pin."""
    from concurrent.futures import ThreadPoolExecutor

    g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    rec = json.load(open(os.path.join(g, "ref_task_config3.json")))
    track = os.path.join(g, "circle_track_300s.csv")
    assert hashlib.md5(open(track, "rb").read()).hexdigest() == rec["track_md5"]
    shm = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else str(tmp_path)
    out = os.path.join(shm, "galtest_%d_config3.ishort" % os.getpid())

    def front_end_oracle():
        rows = pkg.Scenario(NAV, start=rec["start"], duration_s=rec["duration_s"], iono_enable=True, motion_file=track).all()
        static = pkg.Scenario(NAV, llh=(-6, 51, 100), start=rec["start"], duration_s=rec["duration_s"], iono_enable=True).all()
        return _oracle_md5(pkg, rows), rows, static

    with ThreadPoolExecutor(1) as ex:  # (the oracle's 20 s of CPU beside the CLI run)
        want = ex.submit(front_end_oracle)
        try:
            r = subprocess.run([CLI, "-e", NAV, "-u", track, "-t", rec["start"], "-d", str(rec["duration_s"]), "-P", "0", "-o", out],
                               capture_output=True, text=True, timeout=900)
            assert r.returncode == 0, r.stderr
            h, n = hashlib.md5(), 0
            with open(out, "rb") as fh:
                for blk in iter(lambda: fh.read(1 << 24), b""):
                    h.update(blk)
                    n += len(blk)
        finally:
            if os.path.exists(out):
                os.remove(out)
        (md5, m), rows, static = want.result()
    assert rows.shape == (2999, 16) and n == m == 2999 * 260000 * 4
    act = rows["prn"][0] > 0
    assert np.abs((rows["f_carr"] - static["f_carr"])[:, act]).max() > 10.0  # the receiver does move
    assert h.hexdigest() == md5 == rec["front_end_oracle_md5"], "CLI file vs front-end -> oracle (here, and where the fixture was made)"
    assert rec["reference_ignores_u"] is True and rec["reference_md5_with_u"] == rec["static_default_site_md5"] != md5


@pytest.mark.gpu
def test_cli_file_sink_variants_agree(tmp_path):
    """The mapped multi-writer sink (regular files), the sequential sink (GAL_SINK=stream) and stdout give the same bytes;
    an interrupted mapped run is cut to what was written (test_cli_sigint... covers the paced, streaming case)."""
    common = [CLI, "-e", NAV, "-l", "-6,51,100", "-t", "2022/02/20,12:00:00", "-d", "10", "-I", "1", "-P", "0", "-B", "40"]
    a, b, c = tmp_path / "a.ishort", tmp_path / "b.ishort", tmp_path / "c.ishort"
    assert subprocess.run(common + ["-o", str(a), "--writers", "5"], capture_output=True).returncode == 0
    assert subprocess.run(common + ["-o", str(b)], capture_output=True, env=dict(os.environ, GAL_SINK="stream")).returncode == 0
    assert subprocess.run(common + ["-o", str(c), "--writers", "0"], capture_output=True).returncode == 0
    for f in (a, b, c):
        data = f.read_bytes()
        assert len(data) == REF["G1"]["bytes"] and hashlib.md5(data).hexdigest() == REF["G1"]["md5"], f


@pytest.mark.gpu
def test_cli_120s_file_equals_library(pkg, tmp_path):
    """BASELINE configs[0/1] literally -- the reference command line with -d 120 into a file (1199 epochs, 1 246 960 000 bytes:
    src/galileo-sdr.cpp:438,542,658) -- against the same scenario through the library (front-end rows -> gal_synth_run_host);
    bench.py's e2e.file_sink leg times this command and reports the same comparison (md5_equals_library)."""
    shm = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else str(tmp_path)
    out = os.path.join(shm, "galtest_%d_120s.ishort" % os.getpid())
    try:
        r = subprocess.run([CLI, "-e", NAV, "-l", "-6,51,100", "-t", "2022/02/20,12:00:00", "-d", "120", "-U", "1", "-b", "1", "-P", "0",
                            "-o", out], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr
        assert os.path.getsize(out) == 1199 * 260000 * 4
        h = hashlib.md5()
        with open(out, "rb") as fh:
            for blk in iter(lambda: fh.read(1 << 24), b""):
                h.update(blk)
    finally:
        if os.path.exists(out):
            os.remove(out)
    rows = pkg.Scenario(NAV, llh=(-6.0, 51.0, 100.0), start="2022/02/20,12:00:00", duration_s=120.0, iono_enable=True).all()
    assert rows.shape == (1199, 16)
    with pkg.SynthEngine(device=0) as eng:
        iq, _, stats = eng.run_host(rows)
    assert stats["chain_mismatch"] == 0 and stats["kernel_family"] == 1
    assert h.hexdigest() == hashlib.md5(iq.tobytes()).hexdigest()


@pytest.mark.gpu
def test_cli_exact_replay_flag_gives_the_same_bytes(tmp_path):
    """--exact-replay (GAL_CFG_EXACT_REPLAY): the exact-replay kernel instead of the default one of this geometry; G1's md5 either way."""
    out = tmp_path / "g1x.ishort"
    r = subprocess.run([CLI, "-e", NAV, "-l", "-6,51,100", "-t", "2022/02/20,12:00:00", "-d", "10", "-I", "1", "-P", "0", "--exact-replay",
                        "-o", str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert hashlib.md5(out.read_bytes()).hexdigest() == REF["G1"]["md5"]


def test_cli_duration_too_short_leaves_an_empty_file(tmp_path):
    """(int)(10 d + 0.5) < 2 epochs: the reference opens its sink and has nothing to generate (src/galileo-sdr.cpp:438; seen with
    oracle/_ref/ref_task: an empty file); the CLI does the same before it touches the device."""
    out = tmp_path / "e.ishort"
    r = subprocess.run([CLI, "-e", NAV, "-l", "-6,51,100", "-t", "2022/02/20,12:00:00", "-d", "0.14", "-P", "0", "-o", str(out)],
                       capture_output=True, text=True)
    assert r.returncode == 0 and out.exists() and out.stat().st_size == 0, r.stderr

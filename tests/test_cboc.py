"""CBOC(6,1,1/11) opt-in mode (GAL_CFG_CBOC / CLI -C).  `north_star` names the composite sub-carrier of the E1 OS ICD;
the reference generates BOC(1,1) only (src/gal-sig.cpp:198-233), so there is no reference behaviour to be identical to:
the mode is DEFINED by the oracle's CBOC loop (oracle/galsyn_oracle.c), which is checked here against an independent
numpy evaluation of the ICD formula, and the GPU path is bit-exact against that oracle."""
import numpy as np
import pytest

from oracle_binding import oracle_cboc_tables, oracle_run, oracle_tables


def _numpy_cboc(pkg, p, n, rate):
    """e_B (alpha sc_A + beta sc_B) - e_C (alpha sc_A - beta sc_B) per sample, straight from the ICD: chips from the
    memory codes, sc_A / sc_B = -1 in the first half of their period (the reference's sboc convention), integer carrier
    tables lround(alpha LUT), lround(beta LUT); NCO recurrences as in SURVEY.md Appendix B."""
    t = pkg.tables()
    cos, sin, cs25 = t["cos512"].astype(np.float64), t["sin512"].astype(np.float64), t["cs25"]
    alpha, beta = np.sqrt(10.0 / 11.0), np.sqrt(1.0 / 11.0)
    rnd = lambda v: np.where(v >= 0, np.floor(v + 0.5), -np.floor(-v + 0.5)).astype(int)  # lround
    cA, sA, cB, sB = rnd(alpha * cos), rnd(alpha * sin), rnd(beta * cos), rnd(beta * sin)
    delt = 1.0 / rate
    E, S = p.shape
    out = np.zeros(E * n * 2, dtype=np.int64)
    st = {}
    for e in range(E):
        for k in range(n):
            I = Q = 0
            for j in range(S):
                r = p[e, j]
                if r["prn"] <= 0:
                    continue
                if k == 0:
                    if r["flags"] & 1:
                        st[j] = dict(cp=float(r["carr_phase0"]), page=pkg.unpack_page(r["page_init"]))
                    st[j].update(x=float(r["code_phase0"]), ib=int(r["ibit0"]))
                s = st[j]
                if s["x"] >= 4092.0:
                    s["x"] -= 4092.0
                    s["ib"] += 1
                    if s["ib"] >= 500:
                        s["ib"] = 0
                        s["page"] = pkg.unpack_page(r["page_next"])
                kk = int(511 * s["cp"]) & 511
                chip = int(s["x"])
                eb = -1 if (t["e1b"][r["prn"] - 1][chip >> 5] >> (chip & 31)) & 1 else 1
                ec = -1 if (t["e1c"][r["prn"] - 1][chip >> 5] >> (chip & 31)) & 1 else 1
                sc_a = 1 if int(s["x"] * 2) & 1 else -1
                sc_b = 1 if int(s["x"] * 12.0) & 1 else -1
                d = -1 if s["page"][s["ib"]] > 0 else 1
                sec = -1 if (cs25 >> (s["ib"] % 25)) & 1 else 1
                B, C = eb * d, ec * sec
                I += sc_a * (B - C) * cA[kk] + sc_b * (B + C) * cB[kk]
                Q += sc_a * (B - C) * sA[kk] + sc_b * (B + C) * sB[kk]
                s["x"] = s["x"] + float(r["f_code"]) * delt
                s["cp"] = s["cp"] + float(r["f_carr"]) * delt
                s["cp"] = s["cp"] - int(s["cp"])
            out[2 * (e * n + k)] = I
            out[2 * (e * n + k) + 1] = Q
    return out.astype(np.int16)


def test_oracle_cboc_tables():
    cos, sin, _ = oracle_tables()
    cA, sA, cB, sB = oracle_cboc_tables()
    a, b = np.sqrt(10.0 / 11.0), np.sqrt(1.0 / 11.0)
    assert np.abs(cA - a * cos).max() <= 0.5 and np.abs(sB - b * sin).max() <= 0.5
    assert cA.max() == 238 and cB.max() == 75  # 250 alpha, 250 beta
    # power split 10/11 : 1/11 up to the rounding of the tables
    assert abs((cA.astype(float) ** 2).sum() / (cB.astype(float) ** 2).sum() - 10.0) < 0.1


def test_oracle_cboc_against_numpy_icd_formula(pkg):
    n, rate = 600, 25e6  # 25 MS/s resolves the BOC(6,1) component (12.276 M half periods per second)
    p = pkg.workloads.make_synthetic(n_epochs=2, n_chan=3, n_slots=4, samples_per_epoch=n, sample_rate=rate, seed=15)
    p["code_phase0"][0, 0] = 4091.99  # a code wrap + symbol advance ...
    p["ibit0"][0, 0] = 499             # ... that also flips the page
    got, _ = oracle_run(p, n, rate, cboc=True)
    assert np.array_equal(got, _numpy_cboc(pkg, p, n, rate))
    n, rate = 500, 2.6e6               # the reference's rate (the BOC(6,1) component is aliased there, but defined)
    p = pkg.workloads.make_synthetic(n_epochs=2, n_chan=4, n_slots=4, samples_per_epoch=n, sample_rate=rate, seed=16)
    got, _ = oracle_run(p, n, rate, cboc=True)
    assert np.array_equal(got, _numpy_cboc(pkg, p, n, rate))
    # it is a different signal from the reference's BOC(1,1), with the same NCO state
    boc, st_b = oracle_run(p, n, rate)
    _, st_c = oracle_run(p, n, rate, cboc=True)
    assert not np.array_equal(got, boc)
    assert np.array_equal(st_b["carr_phase"].view(np.uint64), st_c["carr_phase"].view(np.uint64))


def _compare_cboc(pkg, p, n, rate=2.6e6, state_in=None, window_mode=None, **eng_kw):
    flags = pkg.synth.GAL_CFG_CBOC
    with pkg.SynthEngine(sample_rate=rate, samples_per_epoch=n, n_slots=p.shape[1], device=0, flags=flags, **eng_kw) as eng:
        iq, st, stats = eng.run_host(p, state_in)
    if window_mode is not None:
        assert stats["window_mode"] == window_mode, stats
    ref_iq, ref_st = oracle_run(p, n, rate, state_in, cboc=True)
    assert stats["chain_mismatch"] == 0
    nbad = int(np.count_nonzero(iq != ref_iq))
    assert nbad == 0, "%d of %d int16 values differ (first at %d)" % (nbad, iq.size, int(np.flatnonzero(iq != ref_iq)[0]))
    act = ref_st["prn"] > 0
    assert np.array_equal(st["carr_phase"][act].view(np.uint64), ref_st["carr_phase"][act].view(np.uint64))
    assert np.array_equal(st["page"][act], ref_st["page"][act])
    return iq, st


@pytest.mark.gpu
@pytest.mark.parametrize("n_chan", [1, 5, 9, 12])
def test_cboc_hip_equals_oracle(pkg, n_chan):
    """At the reference's 2.6 MS/s the mode runs on resampled windows (window_mode 1: chip holds AND the parity of the BOC(6,1)
    half period looked up once per 16-sample group); the classic per-sample body on the same batch gives the same bits."""
    p = pkg.workloads.make_synthetic(n_epochs=3, n_chan=n_chan, n_slots=16, samples_per_epoch=52000, seed=200 + n_chan)
    a, _ = _compare_cboc(pkg, p, 52000, window_mode=1)


@pytest.mark.gpu
def test_cboc_both_bodies_on_the_same_batch(pkg, monkeypatch):
    p = pkg.workloads.make_synthetic(n_epochs=3, n_chan=11, n_slots=16, samples_per_epoch=104000, seed=321)
    a, _ = _compare_cboc(pkg, p, 104000, window_mode=1, test_hooks=True)
    monkeypatch.setenv("GAL_SYNTH_RW", "0")  # honoured by the GAL_TEST_HOOKS build only: classic windows
    b, _ = _compare_cboc(pkg, p, 104000, window_mode=0, test_hooks=True)
    assert np.array_equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("rate,mode", [(2.2e6, 1), (2.4e6, 1), (2.6e6, 1), (2.76e6, 1), (2.1e6, 0), (2.5e6, 0), (3.0e6, 0), (5.0e6, 0),
                                       (40e6, 0), (3.2e6, 4), (4.0e6, 4), (5.5e6, 4), (6.0e6, 4), (8e6, 3), (10e6, 3), (12.5e6, 3),
                                       (16e6, 2), (20e6, 2), (25e6, 2), (30e6, 2)])
def test_cboc_resampled_window_gate_and_rates(pkg, rate, mode):
    """The host enables the resampled-window body only where BOTH threshold sets (chip holds: step s; half-period parity:
    step 6 s) keep more than a bin (1/64) between neighbours; elsewhere the classic body runs.  Same bits everywhere.
    Round 5: the mode runs on every window form of k_synth_g (rounds 3-4: the hold form of the reference's rate only) --
    the parity of a sample's half chip comes out of the hold masks (forms 1, 4) or the advance masks (2, 3)."""
    n = int(rate / 50)
    p = pkg.workloads.make_synthetic(n_epochs=3, n_chan=7, n_slots=8, samples_per_epoch=n, sample_rate=rate, seed=int(rate) % 977)
    p["ibit0"][0, 0] = 499
    p["code_phase0"][0, 1] = 4091.9
    _compare_cboc(pkg, p, n, rate, window_mode=mode)


@pytest.mark.gpu
@pytest.mark.parametrize("rate", [2.1e6, 2.5e6, 3.0e6, 2.0462e6])
def test_cboc_resampled_window_safety_nets(pkg, monkeypatch, rate):
    """The body forced onto rates its gate keeps away (GAL_SYNTH_RW=11): clustered thresholds leave bins undecidable and
    those groups take the per-sample body, a pattern with more than four holds turns the block over to it -- still exact."""
    monkeypatch.setenv("GAL_SYNTH_RW", "11")
    n = int(rate / 50)
    p = pkg.workloads.make_synthetic(n_epochs=2, n_chan=6, n_slots=8, samples_per_epoch=n, sample_rate=rate, seed=5)
    _compare_cboc(pkg, p, n, rate, window_mode=1, test_hooks=True)


@pytest.mark.gpu
def test_cboc_reference_geometry_page_flips_and_split_run(pkg):
    n = 260000
    p = pkg.workloads.make_synthetic(n_epochs=4, n_chan=10, n_slots=16, samples_per_epoch=n, seed=31)
    p["ibit0"][0, :4] = [499, 498, 476, 0]
    for e in range(1, 4):
        p["ibit0"][e, :4] = (p["ibit0"][0, :4] + 25 * e) % 500
    f = np.array([-3400.0, -3.0, 0.03, 3499.0])
    for e in range(4):
        p["f_carr"][e, 4:8] = f + 0.01 * e * np.sign(f)
        p["f_code"][e, 4:8] = 1.023e6 + p["f_carr"][e, 4:8] * 0.0006493506493506494
    iq, st = _compare_cboc(pkg, p, n)
    with pkg.SynthEngine(samples_per_epoch=n, n_slots=16, device=0, flags=pkg.synth.GAL_CFG_CBOC) as eng:
        a, st_a, _ = eng.run_host(p[:2])
        q = p[2:].copy()
        q["flags"][0, :10] = 0
        b, st_b, _ = eng.run_host(q, st_a)
    assert np.array_equal(np.concatenate([a, b]), iq)


@pytest.mark.gpu
def test_cboc_24_channels_at_25_msps(pkg):
    """Two 12-channel launches (the second continues from the stored samples), a rate that resolves BOC(6,1)."""
    n, rate = 250000, 25e6
    p = pkg.workloads.make_synthetic(n_epochs=2, n_chan=24, n_slots=24, samples_per_epoch=n, sample_rate=rate, seed=77)
    _compare_cboc(pkg, p, n, rate)


@pytest.mark.gpu
def test_cboc_cli_flag(pkg, tmp_path):
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    nav = os.path.join(root, "tests", "golden", "20feb2022.rnx")
    out = tmp_path / "cboc.ishort"
    r = subprocess.run([os.path.join(root, "galileo-sdr-sim_amd", "galileo-sdr-sim"), "-e", nav, "-l", "-6,51,100", "-t",
                        "2022/02/20,12:00:00", "-d", "3", "-C", "-P", "0", "-o", str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    rows = pkg.Scenario(nav, llh=(-6, 51, 100), start="2022/02/20,12:00:00", duration_s=3, iono_enable=True).all()
    ref_iq, _ = oracle_run(rows, 260000, 2.6e6, cboc=True)
    assert np.array_equal(np.fromfile(str(out), dtype=np.int16), ref_iq)

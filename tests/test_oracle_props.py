"""Properties of the oracle (CPU restatement of reference src/galileo-sdr.cpp:481-539) that do not need
the reference binary: superposition over channels (integer accumulate), epoch independence of the code
chain, state carry across a split run, and agreement with an independent pure-numpy evaluation of
Appendix B of SURVEY.md on a small case."""
import numpy as np

from oracle_binding import oracle_run


def test_superposition(pkg):
    n = 5200
    p = pkg.workloads.make_synthetic(n_epochs=3, n_chan=5, n_slots=8, samples_per_epoch=n, seed=3)
    full, _ = oracle_run(p, n, 2.6e6)
    acc = np.zeros_like(full, dtype=np.int32)
    for j in range(5):
        q = np.zeros_like(p)
        q[:, j] = p[:, j]
        one, _ = oracle_run(q, n, 2.6e6)
        acc += one
    assert np.array_equal(acc, full.astype(np.int32))


def test_split_run_carries_state(pkg):
    n = 5200
    p = pkg.workloads.make_synthetic(n_epochs=6, n_chan=4, n_slots=8, samples_per_epoch=n, seed=4)
    full, st_full = oracle_run(p, n, 2.6e6)
    a, st_a = oracle_run(p[:2], n, 2.6e6)
    b, st_b = oracle_run(p[2:], n, 2.6e6, st_a)
    assert np.array_equal(np.concatenate([a, b]), full)
    assert np.array_equal(st_b["carr_phase"].view(np.uint64), st_full["carr_phase"].view(np.uint64))


def test_against_numpy_appendix_b(pkg):
    """Independent restatement: plain Python/numpy loop following SURVEY.md Appendix B."""
    n = 700
    p = pkg.workloads.make_synthetic(n_epochs=2, n_chan=3, n_slots=4, samples_per_epoch=n, seed=5)
    p["code_phase0"][0, 0] = 4091.9  # force a code wrap + symbol advance
    p["ibit0"][0, 0] = 499            # ... that also flips the page
    t = pkg.tables()
    cos, sin, cs25 = t["cos512"].astype(int), t["sin512"].astype(int), t["cs25"]
    delt = 1.0 / 2.6e6
    out = np.zeros(2 * 2 * n, dtype=np.int16)
    st = {}
    for e in range(2):
        for k in range(n):
            I = Q = 0
            for j in range(3):
                r = p[e, j]
                if k == 0:
                    if e == 0:
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
                ic = int(s["x"] * 2)
                prn = int(r["prn"])
                bB = (int(t["e1b"][prn - 1][ic >> 6]) >> ((ic >> 1) & 31)) & 1
                bC = (int(t["e1c"][prn - 1][ic >> 6]) >> ((ic >> 1) & 31)) & 1
                half = 1 if ic & 1 else -1
                eB = half * (-1 if bB else 1)
                eC = half * (-1 if bC else 1)
                dsg = -1 if s["page"][s["ib"]] > 0 else 1
                ssg = -1 if (cs25 >> (s["ib"] % 25)) & 1 else 1
                v = eB * dsg - eC * ssg
                I += v * cos[kk]
                Q += v * sin[kk]
                s["x"] = s["x"] + float(r["f_code"]) * delt
                s["cp"] = s["cp"] + float(r["f_carr"]) * delt
                s["cp"] = s["cp"] - float(int(s["cp"]))
            out[2 * (e * n + k)] = I
            out[2 * (e * n + k) + 1] = Q
    ref, _ = oracle_run(p[:, :4], n, 2.6e6)
    assert np.array_equal(out, ref)

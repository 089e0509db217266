"""csrc/nco_walk.h on the HOST (libgalwalk_host.so, g++) against brute-force stepping of the reference
recurrences (src/galileo-sdr.cpp:491-507, 528-532).  The product runs the same header on the GPU; these
tests pin the closed forms, including the rounding-tie and sign-change corner cases, bit for bit."""
import ctypes
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBW = os.path.join(ROOT, "galileo-sdr-sim_amd", "libgalwalk_host.so")
DELT = 1.0 / 2600000.0


@pytest.fixture(scope="module")
def W(pkg):
    if not os.path.exists(LIBW):
        pkg.build_all(targets=("libgalwalk_host.so",))
    lib = ctypes.CDLL(LIBW)
    d, i, vp = ctypes.c_double, ctypes.c_int, ctypes.c_void_p
    lib.galwalk_carr.restype = d
    lib.galwalk_carr.argtypes = [d, d, i, i, vp, vp]
    lib.galwalk_carr_brute.restype = d
    lib.galwalk_carr_brute.argtypes = [d, d, i, i, vp]
    lib.galwalk_carr_iters.restype = ctypes.c_long
    lib.galwalk_carr_iters.argtypes = [d, d, i]
    lib.galwalk_code.argtypes = [d, i, d, i, i, vp, vp, vp, vp, vp]
    lib.galwalk_code_brute.argtypes = [d, i, d, i, i, vp, vp, vp, vp, vp]
    lib.galwalk_code_legs.argtypes = [d, i, d, i, i, i, vp, vp, vp, vp, vp, vp, i]
    lib.galwalk_spec_wrap.restype = i
    lib.galwalk_spec_wrap.argtypes = [i, i, i, i, vp, vp, vp, vp, d, i, vp, vp, vp, i, i, vp, i, vp]
    return lib


def _carr_pair(W, p, d, N, R):
    nc = (N + R - 1) // R
    a, b = np.zeros(nc), np.zeros(nc)
    e1 = W.galwalk_carr(p, d, N, R, a.ctypes.data, None)
    e2 = W.galwalk_carr_brute(p, d, N, R, b.ctypes.data)
    return e1, e2, a, b


def _assert_carr(W, p, d, N=260000, R=1016):
    e1, e2, a, b = _carr_pair(W, p, d, N, R)
    assert np.float64(e1).view(np.uint64) == np.float64(e2).view(np.uint64), (p, d, e1, e2)
    assert np.array_equal(a.view(np.uint64), b.view(np.uint64)), (p, d)


def test_carrier_random(W):
    rng = np.random.default_rng(1)
    for t in range(200):
        f = rng.uniform(-4000, 4000) if t % 3 else rng.uniform(-30, 30)
        p = rng.uniform(-1, 1) if t % 2 else rng.uniform(0, 1)
        _assert_carr(W, p, f * DELT)


def test_carrier_rounding_ties(W):
    """Steps that are exact half-ulps of a visited binade (ties to even) and steps with very few bits."""
    rng = np.random.default_rng(2)
    for k in range(40, 54):  # d = odd multiple of 2^-k
        for _ in range(6):
            m = int(rng.integers(1, 1 << 12)) | 1
            d = np.ldexp(float(m), -k)
            d = d * (0.0013 / d) if False else np.ldexp(float(int(0.0013 * 2.0**k) | 1), -k)
            for sgn in (1.0, -1.0):
                _assert_carr(W, rng.uniform(0, 1), sgn * d, N=60000, R=500)
                _assert_carr(W, -rng.uniform(0, 1), sgn * d, N=60000, R=500)


def test_carrier_edge_values(W):
    for p, d in [(0.0, 1e-3), (0.0, -1e-3), (0.999999, 1e-9), (-0.999999, -1e-9), (0.5, 2.0**-60), (0.25, -(2.0**-70)),
                 (0.3, 0.0), (1e-300, 1e-3), (0.75, 0.49), (-0.75, -0.49), (0.1, -0.3), (2.0**-30, 2.0**-29)]:
        _assert_carr(W, p, d, N=20000, R=64)


def test_carrier_iteration_budget(W):
    """Closed form must stay O(binade crossings), not O(samples): regression guard for the batching."""
    rng = np.random.default_rng(3)
    for _ in range(50):
        f = rng.uniform(-3500, 3500)
        it = W.galwalk_carr_iters(rng.uniform(0, 1), f * DELT, 260000)
        assert it < 40 * (abs(f) * 0.1 + 2) + 200, (f, it)


def _code_pair(W, x, ib, c, N, R):
    nc = (N + R - 1) // R
    out = []
    for fn in (W.galwalk_code, W.galwalk_code_brute):
        cx, ci = np.zeros(nc), np.zeros(nc, dtype=np.uint32)
        xe, ie, fl = ctypes.c_double(), ctypes.c_int(), ctypes.c_int()
        fn(x, ib, c, N, R, cx.ctypes.data, ci.ctypes.data, ctypes.byref(xe), ctypes.byref(ie), ctypes.byref(fl))
        out.append((cx, ci, xe.value, ie.value, fl.value))
    return out


def test_code_random_and_flips(W):
    rng = np.random.default_rng(4)
    for t in range(120):
        f = rng.uniform(-3500, 3500)
        c = (1.023e6 + f * 0.0006493506493506494) * DELT
        x = rng.uniform(0, 4092)
        ib = int(rng.integers(0, 500)) if t % 4 else int(rng.integers(470, 500))
        (ax, ai, axe, aie, afl), (bx, bi, bxe, bie, bfl) = _code_pair(W, x, ib, c, 260000, 1016)
        assert np.array_equal(ax.view(np.uint64), bx.view(np.uint64)) and np.array_equal(ai, bi)
        assert (np.float64(axe).view(np.uint64), aie, afl) == (np.float64(bxe).view(np.uint64), bie, bfl)


def test_code_other_rates_and_edges(W):
    for rate, N in [(25e6, 250000), (2.0e6, 50000), (4.092e6, 40000), (1.0e6, 30000)]:
        c = 1.023e6 / rate
        for x, ib in [(0.0, 0), (4091.999, 499), (4092.0, 499), (2047.5, 250), (np.nextafter(4092.0, 0), 10)]:
            (ax, ai, axe, aie, afl), (bx, bi, bxe, bie, bfl) = _code_pair(W, x, ib, c, N, 256)
            assert np.array_equal(ax.view(np.uint64), bx.view(np.uint64)) and np.array_equal(ai, bi)
            assert (np.float64(axe).view(np.uint64), aie, afl) == (np.float64(bxe).view(np.uint64), bie, bfl)


def test_code_small_steps_are_stepped_one_by_one(W):
    """Steps below what gal_synth_plan admits (f_code / fs < 2^-20): the closed-form batches of code_walk estimate their quotient with
    the reciprocal of the STEP, which can overshoot floor(t / dk) by more than the one the remainder test takes back once the step
    is within ~2^20 ulps of the phase (16 of 3000 cases with steps 1e-11 .. 2e-9 near binade tops and the wrap) -- such
    steps are taken one by one; at and above 2^-20 the batches are exact (steps 2^-20 .. 2^-12 here, the rates elsewhere)."""
    rng = np.random.default_rng(9)
    for t in range(400):
        c = float(10.0 ** rng.uniform(-11.5, -6.5)) if t % 2 else float(2.0 ** rng.uniform(-20.0, -12.0))
        k = int(rng.integers(1, 12))
        top = 2.0 ** k if t % 3 else 4092.0
        x = top - c * float(rng.integers(0, 40000)) - (rng.random() < 0.5) * 2.0 ** (k - 52) * float(rng.integers(0, 8))
        x = min(max(x, 0.0), 4092.0)
        (ax, ai, axe, aie, afl), (bx, bi, bxe, bie, bfl) = _code_pair(W, x, int(rng.integers(0, 500)), c, 60000, 1024)
        assert np.array_equal(ax.view(np.uint64), bx.view(np.uint64)) and np.array_equal(ai, bi), (x, c)
        assert (np.float64(axe).view(np.uint64), aie, afl) == (np.float64(bxe).view(np.uint64), bie, bfl), (x, c)


def _code_legs(W, x, ib, c, N, R, legs, force_tie=0):
    nc = (N + R - 1) // R
    cx, ci = np.zeros(nc), np.zeros(nc, dtype=np.uint32)
    xe, ie, fl = ctypes.c_double(), ctypes.c_int(), ctypes.c_int()
    st = np.zeros(4, dtype=np.int32)
    W.galwalk_code_legs(x, ib, c, N, R, legs, cx.ctypes.data, ci.ctypes.data, ctypes.byref(xe), ctypes.byref(ie), ctypes.byref(fl),
                        st.ctypes.data, force_tie)
    bx, bi = np.zeros(nc), np.zeros(nc, dtype=np.uint32)
    bxe, bie, bfl = ctypes.c_double(), ctypes.c_int(), ctypes.c_int()
    W.galwalk_code_brute(x, ib, c, N, R, bx.ctypes.data, bi.ctypes.data, ctypes.byref(bxe), ctypes.byref(bie), ctypes.byref(bfl))
    ok = (np.array_equal(cx.view(np.uint64), bx.view(np.uint64)) and np.array_equal(ci, bi) and
          (np.float64(xe.value).view(np.uint64), ie.value, fl.value) == (np.float64(bxe.value).view(np.uint64), bie.value, bfl.value))
    return ok, st


def test_code_chain_in_legs(W):
    """The code chain of an epoch in speculative legs (nco_walk.h: code_leg_walk / code_ideal_anchor / code_leg_accept, the host
    statement of k_walk_code's lanes and their in-wave stitch): every leg walked from its ideal-arithmetic anchor, then accepted as
    walked, translated by the anchor's error, or walked again -- checkpoints, symbol counters, flip flags and end state must be
    brute-force stepping's, bit for bit, whatever the stitch decided; at the reference's geometry nearly every leg is translated."""
    rng = np.random.default_rng(41)
    tot = np.zeros(4, dtype=np.int64)
    for t in range(160):
        f = rng.uniform(-6000, 6000)
        c = (1.023e6 + f * 0.0006493506493506494) * DELT
        x = rng.uniform(0, 4092) if t % 5 else rng.uniform(4092, 6138)  # (a wrap pending at the epoch's first sample)
        ib = int(rng.integers(0, 500)) if t % 4 else int(rng.integers(470, 500))
        legs = int(rng.choice([2, 4, 8, 16]))
        N, R = (260000, 1024) if t % 3 else (int(rng.integers(3000, 90000)), int(rng.choice([256, 416, 1024, 1040])))
        ok, st = _code_legs(W, x, ib, c, N, R, legs)
        assert ok, (x, ib, c, N, R, legs)
        tot += st
    assert tot[1] > 4 * (tot[0] + tot[2]), tot  # translation is the rule
    # other sample rates (steps 0.04 .. 1 chip per sample), few-bit steps (every binade somebody's tie binade), tie-prone steps
    for rate, N in [(25e6, 250000), (2.0e6, 50000), (4.092e6, 40000), (1.023e6, 30000), (8e6, 80000)]:
        for k in range(12):
            c = 1.023e6 / rate * (1.0 + rng.uniform(-3e-6, 3e-6))
            if k % 3 == 0:
                c = float(np.round(c * 2.0 ** (20 + 3 * (k // 3))) / 2.0 ** (20 + 3 * (k // 3)))  # a multiple of 2^-20 .. 2^-29: tie-prone
            ok, st = _code_legs(W, rng.uniform(0, 4092), int(rng.integers(0, 500)), c, N, 256, int(rng.choice([2, 4, 8])))
            assert ok, (rate, c)
    # steps that are ODD multiples of 2^-42 at the reference's geometry: every addition in [2048, 4096) is a tie, a shift by an odd
    # multiple of 2^-41 flips its resolution once -- translated across it (code_leg_accept: 3).  (Seldom needed: behind the
    # first such addition every residual is an even multiple, and the guesses, computed near 1e5, are multiples of 2^-36.)
    tie = np.zeros(4, dtype=np.int64)
    for t in range(200):
        c = (1.023e6 + rng.uniform(-6000, 6000) * 0.0006493506493506494) * DELT
        c = (np.floor(c * 2.0 ** 42) // 2 * 2 + 1) / 2.0 ** 42
        ok, st = _code_legs(W, rng.uniform(0, 6000), int(rng.integers(0, 500)), float(c), 260000 if t % 2 else int(rng.integers(20000, 90000)),
                            1024, int(rng.choice([2, 4, 8, 16])))
        assert ok, c
        tie += st
    assert tie[3] >= 5 and tie[1] > 100, tie
    # every leg walked again (forced): the serial fallback is exact too
    for t in range(20):
        c = (1.023e6 + rng.uniform(-3500, 3500) * 0.0006493506493506494) * DELT
        ok, st = _code_legs(W, rng.uniform(0, 4092), int(rng.integers(0, 500)), c, 100000, 1024, 4, force_tie=1)
        assert ok and st[1] == 0, st


def test_code_leg_translation_respects_its_margin(W):
    """States within a few ulps of a binade boundary or of the wrap threshold: the margin must send such legs back to be walked --
    phases placed so that the second leg's first wrap residual is tiny (a shift the size of the anchor's error would un-wrap it) and
    so that a crossing into [2048, 4096) lands within 2^-38 of 2048."""
    rng = np.random.default_rng(43)
    tot = np.zeros(4, dtype=np.int64)
    for t in range(600):
        c = (1.023e6 + rng.uniform(-3500, 3500) * 0.0006493506493506494) * DELT
        k = int(rng.integers(15400, 59000))  # (behind the first leg: in a leg that is walked from a guessed anchor)
        # x0 such that the ideal phase k samples on sits a hair beside a wrap threshold (t even) or beside 2048 (t odd)
        target = (4092.0 * (1 + (t % 6) // 2) if t % 2 == 0 else 2048.0) + float(rng.integers(-6, 7)) * 2.0 ** -39
        x = (target - k * c) % 4092.0
        ok, st = _code_legs(W, x, int(rng.integers(0, 500)), c, 60000, 1024, 4)
        assert ok, (x, c, k)
        tot += st
    assert tot[2] > 100 and tot[1] > 100, tot  # both outcomes seen: sent back to be walked, and translated


def _chain_truth(W, p, d, N):
    ends = np.zeros(len(d))
    for e in range(len(d)):
        p = W.galwalk_carr_brute(p, d[e], N, N, None)
        ends[e] = p
    return ends


def test_wrap_anchored_stitching_as_on_gpu(W):
    """The scheme the GPU runs (k_walk_carr / k_scanm): legs anchored at the last wrap, claims stitched by
    the sequential statement (nthreads=0) and by the block-parallel three-sweep form (nthreads=256).  Both must
    reproduce the sequential chain exactly and converge in a handful of passes -- also for channels sweeping
    through zero Doppler, where leg-boundary speculation needed hundreds of passes."""
    rng = np.random.default_rng(7)
    E, N, Wl, L = 80, 26000, 8, 3264
    cases = []
    for t in range(4):
        f0 = rng.uniform(-3000, 3000)
        cases.append((f0 - 0.05 * np.arange(E)) * DELT)
    cases.append(np.linspace(40.0, -35.0, E) * DELT)      # zero crossing
    cases.append(np.full(E, 0.7) * DELT)                   # practically no wraps at all
    cases.append(-(1500.0 + 0.5 * np.arange(E)) * DELT)    # negative Doppler
    for d in cases:
        d = np.ascontiguousarray(d)
        prn = np.full(E, 3, dtype=np.int32)
        flags = np.zeros(E, dtype=np.uint32)
        p0 = np.zeros(E)
        flags[0] = 1
        p0[0] = rng.uniform(0, 1)
        truth = _chain_truth(W, p0[0], d, N)
        passes = []
        for nthreads in (0, 256, 1024):
            pend = np.zeros(E * Wl)
            hist = np.zeros(64, dtype=np.int32)
            walks = ctypes.c_long()
            r = W.galwalk_spec_wrap(E, Wl, L, N, prn.ctypes.data, flags.ctypes.data, p0.ctypes.data, d.ctypes.data,
                                    0.0, 64, pend.ctypes.data, ctypes.byref(walks), hist.ctypes.data, nthreads,
                                    0, None, 0, None)
            assert 0 < r <= 6, (r, nthreads, hist[:12])
            assert np.array_equal(pend[Wl - 1::Wl].view(np.uint64), truth.view(np.uint64))
            passes.append(r)
        # the block-parallel form composes the pending-correction maps exactly: same pass count as sequential
        assert passes[0] == passes[1] == passes[2], passes


def _chain_truth_cp(W, p, d, N, R):
    """Brute-force chain: phase before every R-th sample of every epoch, and the epoch end phases."""
    nc = (N + R - 1) // R
    cps = np.zeros((len(d), nc))
    ends = np.zeros(len(d))
    for e in range(len(d)):
        p = W.galwalk_carr_brute(p, d[e], N, R, cps[e].ctypes.data)
        ends[e] = p
    return cps, ends


def _run_spec(W, d, p_start, N, Wl, L, R, translate, nthreads=256):
    E = len(d)
    prn = np.full(E, 3, dtype=np.int32)
    flags = np.zeros(E, dtype=np.uint32)
    p0 = np.zeros(E)
    flags[0] = 1
    p0[0] = p_start
    pend = np.zeros(E * Wl)
    cp = np.zeros(E * Wl * (L // R))
    walks, shifts = ctypes.c_long(), ctypes.c_long()
    r = W.galwalk_spec_wrap(E, Wl, L, N, prn.ctypes.data, flags.ctypes.data, p0.ctypes.data, d.ctypes.data, 0.0, 64,
                            pend.ctypes.data, ctypes.byref(walks), None, nthreads, R, cp.ctypes.data, translate,
                            ctypes.byref(shifts))
    return r, pend, cp.reshape(E, Wl * (L // R)), walks.value, shifts.value


def test_translated_legs_are_bit_exact(W):
    """Translated acceptance (k_walk_carr dirty == 2): a leg whose anchor residual moved by less than its binade
    margin is shifted instead of walked again.  Every checkpoint and every leg end must still equal the chain
    stepped sample by sample, and most second-pass walks must disappear."""
    rng = np.random.default_rng(11)
    N, Wl, L, R = 26000, 8, 3264, 102
    ncp = (N + R - 1) // R
    cases = []
    for t in range(6):
        f0 = rng.uniform(-3500, 3500)
        cases.append((f0 - 0.05 * np.arange(400)) * DELT)
    cases.append(np.linspace(60.0, -60.0, 300) * DELT)                    # sign change, slow
    cases.append(-(2500.0 + 0.3 * np.arange(300)) * DELT)                 # negative Doppler
    cases.append(np.full(200, 1234.5) * DELT)                             # constant step
    tie = np.full(120, np.ldexp(2 * 6001 + 1, -53))                       # odd multiple of 2^-53: tie-prone
    cases.append(tie)
    mix = (1800.0 + 0.01 * np.arange(200)) * DELT
    mix[50:60] = np.ldexp(2 * 5003 + 1, -53)                              # a few tie-prone epochs inside
    cases.append(mix)
    # steps with few significant bits sprinkled between ordinary ones: multiples of 2^-k around the 2^-52/2^-53
    # grids of the wrap step (an EVEN multiple of 2^-53 still ties at the first wrap of its epoch, because the
    # phase it inherits may carry the 2^-53 bit -- found by the replay check on the M-DYN workload)
    for k in (50, 52, 53, 54, 56):
        dd = (rng.uniform(-3000, 3000) + 0.07 * np.arange(240)) * DELT
        sel = rng.random(240) < 0.25
        dd[sel] = np.round(dd[sel] * 2.0 ** k) / 2.0 ** k
        cases.append(dd)
    tot_w0 = tot_w1 = tot_s = 0
    for ci, d in enumerate(cases):
        d = np.ascontiguousarray(d)
        p_start = rng.uniform(0, 1)
        cps, ends = _chain_truth_cp(W, p_start, d, N, R)
        for translate in (0, 1):
            r, pend, cp, walks, shifts = _run_spec(W, d, p_start, N, Wl, L, R, translate)
            # the tie flip is predicted exactly (WalkOut::tdir): tie-prone epochs cost no extra passes
            assert 0 < r <= 4, (r, translate, ci)
            assert np.array_equal(pend[Wl - 1::Wl].view(np.uint64), ends.view(np.uint64)), translate
            assert np.array_equal(cp[:, :ncp].view(np.uint64), cps.view(np.uint64)), translate
            if translate:
                tot_w1 += walks
                tot_s += shifts
            else:
                tot_w0 += walks
                assert shifts == 0
    assert tot_s > 0 and tot_w1 < 0.7 * tot_w0, (tot_w0, tot_w1, tot_s)


def test_resampled_window_gate(pkg):
    """Host-side gate of k_synth's resampled-window body (synth_api.cpp: rw_threshold_gap): the smallest distance between
    two of the 15 hold-pattern thresholds 1 - frac(u s), against an independent numpy evaluation; and the decisions the
    GPU rate tests rely on (2.6 MS/s qualifies with a wide margin, rates whose code step is near a fraction with a small
    denominator do not).  The GAL_TEST_HOOKS build exports the function; no device is needed."""
    import ctypes

    lib = pkg.synth.load_library(hooks=True)
    lib.gal_hooks_rw_threshold_gap.restype = ctypes.c_double
    lib.gal_hooks_rw_threshold_gap.argtypes = [ctypes.c_double]
    lib.gal_hooks_rw_min_gap.restype = ctypes.c_double
    need = lib.gal_hooks_rw_min_gap()
    assert 1.0 / 128.0 < need < 1.0 / 128.0 + 1e-5

    def gap(s):
        t = np.sort(1.0 - np.mod(np.arange(1, 16) * s, 1.0))
        return float(np.diff(t).min())

    rng = np.random.default_rng(5)
    for s in list(rng.uniform(0.74, 0.9999, 200)) + [2 * 1.023e6 / fs for fs in (2.1e6, 2.2e6, 2.4e6, 2.6e6, 2.76e6)]:
        assert abs(lib.gal_hooks_rw_threshold_gap(float(s)) - gap(s)) < 1e-12, s
    for fs, ok in ((2.6e6, True), (2.1e6, True), (2.76e6, True), (2.5e6, False), (2.728e6, False), (2.0462e6, False)):
        assert (lib.gal_hooks_rw_threshold_gap(2 * 1.023e6 / fs) > need) == ok, fs
    # the whole Doppler range of the reference's rate keeps its distance (f_code = 1.023e6 (1 +- 3.5e3 / 1575.42e6))
    for d in np.linspace(-3500.0, 3500.0, 29):
        assert lib.gal_hooks_rw_threshold_gap(2 * (1.023e6 + d * 0.0006493506493506494) / 2.6e6) > 2 * need

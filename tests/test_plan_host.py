"""The HOST side of gal_synth_plan on its own (GAL_TEST_HOOKS build, no device): what a plan stages for the upload -- the SoA split
of the records, the NCO steps, the compact page_init table, the carrier guesses -- against numpy on the same records, and what it
costs (the engine on FRESH parameters; the reference computes its parameters between epochs,
src/galileo-sdr.cpp:450-479)."""
import ctypes

import numpy as np
import pytest


def _hooks(pkg):
    lib = pkg.synth.load_library(hooks=True)
    lib.gal_hooks_plan_host_ms.restype = ctypes.c_double
    lib.gal_hooks_plan_host_ms.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32]
    lib.gal_hooks_plan_host_array.restype = ctypes.c_void_p
    lib.gal_hooks_plan_host_array.argtypes = [ctypes.c_char_p]
    return lib


def _plan(pkg, lib, p, n_samp=260000, rate=2.6e6, flags=0, reps=1, state=None):
    cfg = pkg.synth._Cfg(float(rate), int(n_samp), int(p.shape[1]), 0, 0, 0, int(flags))
    p = np.ascontiguousarray(p)
    ms = lib.gal_hooks_plan_host_ms(ctypes.byref(cfg), p.ctypes.data, p.shape[0], state.ctypes.data if state is not None else None, reps)
    assert ms >= 0.0
    return ms


def _arr(lib, name, dtype, count):
    ptr = lib.gal_hooks_plan_host_array(name.encode())
    assert ptr
    return np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_uint8)), shape=(count * np.dtype(dtype).itemsize,)).view(dtype).copy()


@pytest.mark.parametrize("dyn", [False, True])
def test_staged_arrays_equal_numpy(pkg, dyn):
    lib = _hooks(pkg)
    E, S = 97, 16
    p = pkg.workloads.make_synthetic(n_epochs=E, n_chan=11, n_slots=S, samples_per_epoch=260000, seed=31, dyn_track=dyn)
    p["flags"][40, 3] |= 1  # a re-allocation mid-batch: its page_init must be in the compact table
    p["carr_phase0"][40, 3] = -0.0
    _plan(pkg, lib, p)
    n = E * S
    flat = p.reshape(-1)
    delt = 1.0 / 2.6e6
    assert np.array_equal(_arr(lib, "prn", np.int32, n), flat["prn"])
    assert np.array_equal(_arr(lib, "flags", np.uint32, n), flat["flags"])
    assert np.array_equal(_arr(lib, "ib0", np.int32, n), flat["ibit0"])
    assert np.array_equal(_arr(lib, "x0", np.float64, n).view(np.uint64), flat["code_phase0"].view(np.uint64))
    p0 = _arr(lib, "p0", np.float64, n)
    assert np.array_equal(p0, flat["carr_phase0"]) and not np.signbit(p0[40 * S + 3])  # -0.0 canonicalised
    # one IEEE multiplication each, the reference's own product (src/galileo-sdr.cpp:528,531)
    assert np.array_equal(_arr(lib, "cstep", np.float64, n).view(np.uint64), (flat["f_code"] * delt).view(np.uint64))
    assert np.array_equal(_arr(lib, "dstep", np.float64, n).view(np.uint64), (flat["f_carr"] * delt).view(np.uint64))
    assert np.array_equal(_arr(lib, "page_next", np.uint32, n * 16).reshape(n, 16), flat["page_next"])
    restart = (flat["prn"] > 0) & ((flat["flags"] & 1) != 0)
    ix = _arr(lib, "init_ix", np.int32, n)
    table = _arr(lib, "page_init", np.uint32, int(restart.sum()) * 16).reshape(-1, 16)
    assert restart.sum() == 12 and np.array_equal(ix[restart], np.arange(restart.sum()))
    assert np.array_equal(table, flat["page_init"][restart])
    # the guesses of active records are finite phases in (-1, 1) and wrap events at or before the epoch start
    act = flat["prn"] > 0
    g = _arr(lib, "pguess", np.float64, n)
    w = _arr(lib, "gss_w", np.int64, n)
    assert np.all(np.abs(g[act]) < 1.0) and np.all(w[act] <= (np.arange(n) // S * 260000)[act]) and np.all(w[act] >= 0)
    assert int(lib.gal_hooks_plan_host_array(b"family")) - 1 == 1  # the reference geometry takes the group kernel


def test_plan_twice_stages_the_same_bytes(pkg):
    """The staging buffer is not cleared between plans (round 6): a plan of other records in between must not leak into the next."""
    lib = _hooks(pkg)
    a = pkg.workloads.make_synthetic(n_epochs=40, n_chan=9, n_slots=16, samples_per_epoch=260000, seed=5)
    b = pkg.workloads.make_synthetic(n_epochs=64, n_chan=12, n_slots=16, samples_per_epoch=260000, seed=6)
    names = (("prn", np.int32, 1), ("x0", np.float64, 1), ("dstep", np.float64, 1), ("pguess", np.float64, 1), ("gss_w", np.int64, 1),
             ("gss_r", np.float64, 1), ("init_ix", np.int32, 1), ("page_next", np.uint32, 16))
    _plan(pkg, lib, a)
    first = [_arr(lib, nm, dt, 40 * 16 * k) for nm, dt, k in names]
    _plan(pkg, lib, b)
    _plan(pkg, lib, a)
    again = [_arr(lib, nm, dt, 40 * 16 * k) for nm, dt, k in names]
    for x, y, (nm, _, _) in zip(first, again, names):
        assert np.array_equal(x.view(np.uint8), y.view(np.uint8)), nm


def test_window_gate_cache_agrees_with_a_fresh_evaluation(pkg):
    """gal_synth_plan evaluates the hold-pattern gates once per slot and reuses the verdict inside a radius in which no threshold can
    cross another, 0 or 1 (rw_rad): over rates where the gate passes, fails and hangs on the edge, the kernel family chosen with a
    warm cache (many epochs, Doppler moving every step) must be the one a one-epoch plan of any of its epochs gets."""
    lib = _hooks(pkg)
    for rate in (2.6e6, 2.5e6, 2.728e6, 3.0e6, 3.8e6, 4.0e6, 4.092e6, 6.5e6, 16e6, 25e6):
        n = int(rate / 10)
        p = pkg.workloads.make_synthetic(n_epochs=60, n_chan=6, n_slots=8, samples_per_epoch=n, sample_rate=rate, seed=77, dyn_track=True)
        def verdict():  # (kernel family, window form, bin tables or bisection)
            return tuple(int(lib.gal_hooks_plan_host_array(k)) - 1 for k in (b"family", b"form", b"search"))

        _plan(pkg, lib, p, n_samp=n, rate=rate)
        fam = verdict()
        fams = set()
        for e in (0, 17, 59):
            q = p[e:e + 1].copy()
            q["flags"][0, :] |= 1
            _plan(pkg, lib, q, n_samp=n, rate=rate)
            fams.add(verdict())
        assert fams == {fam}, (rate, fam, fams)
        assert fam[0] == 1 and fam[2] == (1 if rate in (2.5e6, 2.728e6, 3.8e6, 4.092e6) else 0), (rate, fam)


def test_host_plan_time_is_reported(pkg, capsys):
    """1199 epochs x 16 slots (M-SYN12): the host work of one plan, best of 5 -- printed, and bounded loosely (this container's cores
    are not the GPU box's; the bench line carries the live figure, configs.fresh_plan.plan_ms)."""
    lib = _hooks(pkg)
    p = pkg.shard.rank_workload(0, 1199)
    ms = _plan(pkg, lib, p, reps=5)
    with capsys.disabled():
        print("\n[host plan, 1199 x 16 records: %.3f ms]" % ms)
    assert ms < 20.0


def test_carrier_guesses_through_a_doppler_zero_crossing(pkg):
    """The first guesses of the speculative carrier walk (gal_synth_plan, GuessChain) against the exact chain, epoch by epoch, for
    carriers whose Doppler passes through zero (a satellite at culmination) and for one that flips its sign at 3 kHz: the guessed
    phase at every epoch start is the exact one to 1e-9 cycles IN THE REFERENCE'S REPRESENTATION -- the phase keeps its sign until it
    crosses zero, and crossing zero is not a wrap (src/galileo-sdr.cpp:531-532) -- and the guessed last wrap event in front of an
    epoch is the true one.  (Rounds 1-5 gave the carried phase the sign of the NEXT step: a phantom wrap behind every sign change, a
    second and a third walker pass for the batch: tools/fresh_plan_probe.py.)"""
    from test_walker_cpu import LIBW

    lib = _hooks(pkg)
    w = ctypes.CDLL(LIBW)
    w.galwalk_carr.restype = ctypes.c_double
    w.galwalk_carr.argtypes = [ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    E, S, N, rate = 300, 8, 26000, 2.6e6
    p = pkg.workloads.make_synthetic(n_epochs=E, n_chan=4, n_slots=S, samples_per_epoch=N, sample_rate=rate, seed=9)
    e = np.arange(E)
    p["f_carr"][:, 0] = 30.0 - 0.2 * e          # crosses zero at epoch 150, slowly
    p["f_carr"][:, 1] = -3.0 + 0.02 * e         # ... from below, more slowly still
    p["f_carr"][:, 2] = np.where(e % 50 < 25, 3000.0, -3000.0)  # flips its sign at full speed
    p["f_carr"][:, 3] = -1234.5                 # an ordinary negative Doppler
    p["f_code"][:, :4] = 1.023e6 + p["f_carr"][:, :4] * 0.0006493506493506494
    _plan(pkg, lib, p, n_samp=N, rate=rate)
    n = E * S
    pg = _arr(lib, "pguess", np.float64, n).reshape(E, S)
    gw = _arr(lib, "gss_w", np.int64, n).reshape(E, S)
    gr = _arr(lib, "gss_r", np.float64, n).reshape(E, S)
    dstep = _arr(lib, "dstep", np.float64, n).reshape(E, S)
    for s in range(4):
        ph = float(p["carr_phase0"][0, s])
        last_w, last_r = 0, ph
        cp = np.zeros(N + 1)
        for ep in range(E):
            assert abs(pg[ep, s] - ph) < 1e-9, (s, ep, pg[ep, s], ph)            # same representation, not just the same phase mod 1
            assert gw[ep, s] == last_w and abs(gr[ep, s] - last_r) < 1e-9, (s, ep, gw[ep, s], last_w)
            d = float(dstep[ep, s])
            end = w.galwalk_carr(ph, d, N, 1, cp.ctypes.data, None)           # the exact phase before every sample of the epoch
            cp[N] = end
            wraps = np.flatnonzero(np.abs(cp[:N] + d) >= 1.0)                   # samples whose step wrapped: `p -= (long)p` took something
            if wraps.size:
                last_w, last_r = ep * N + int(wraps[-1]) + 1, float(cp[wraps[-1] + 1])
            ph = end

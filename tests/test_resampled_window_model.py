"""CPU model of k_synth's resampled-window group start (csrc/synth_kernels.hip: rw_phase_a/b/c, GAL_ADV, the table
build in the block prologue), in numpy, against brute-force sequential FP64 stepping of the reference's code NCO
(src/galileo-sdr.cpp:528, `code_phase += f_code * delt`, here in half chips).  It checks the ARGUMENT the kernel
rests on, independent of the GPU:

  * whenever the model does not raise `unsafe`, the hold / advance pattern it looks up reproduces (int)y_u of the
    sequentially summed y_u for all 16 samples of the group -- including fractions placed a few float ulps from a
    threshold, where only the safety margin separates right from wrong;
  * outside the tie binade, and unless the result leaves y's binade, fma(fl(y + s) - y, 16, y) IS the 16th sequential sum;
  * the tie binade is e_s + 1 + ctz(significand(s)), and inside it the shortcut does fail for some y.

The constants mirror the kernel's (RW_BINS, RW_EDGE, RW_DELTA); the GPU tests compare the kernel itself with the oracle."""
import math

import numpy as np
import pytest

BINS = 128
EDGE = np.float32(2.0 ** -20)
DELTA = np.float32(2.0 ** -22)


def build_tables(s, mode):
    """(thr[129], idb[129], patterns[16] as lists of event positions) exactly as the block prologue builds them."""
    u = np.arange(1, 16, dtype=np.float64)
    T = 1.0 - np.mod(u * s, 1.0)
    order = np.lexsort((np.arange(15), T))  # rank by (T, u): ties by the smaller u first
    Ts = T[order].astype(np.float32)
    thr = np.full(BINS + 1, 4.0, dtype=np.float32)
    idb = np.zeros(BINS + 1, dtype=np.int64)
    for b in range(BINS + 1):
        lo = np.float32(b) * np.float32(1.0 / BINS) - EDGE
        hi = np.float32(b + 1) * np.float32(1.0 / BINS) + EDGE
        inside = (Ts >= lo) & (Ts < hi)
        idb[b] = int(np.count_nonzero(Ts < lo))
        if inside.sum() == 1:
            thr[b] = Ts[inside][0]
        elif inside.sum() >= 2 or b == BINS:
            thr[b] = np.float32("nan")
    thr[BINS] = np.float32("nan")
    pats, overflow = [], False
    for idn in range(16):
        tlo = float(Ts[idn - 1]) if idn else 0.0
        thi = min(float(Ts[idn]) if idn < 15 else 2.0, 1.0)
        f = 0.5 * (tlo + thi)
        ev, gp = [], 0.0
        for uu in range(1, 16):
            g = math.floor(f + uu * s)
            if (g != gp) if mode == 2 else (g == gp):
                ev.append(uu)
            gp = g
        overflow |= len(ev) > (2 if mode == 2 else 4)
        pats.append(ev[:4])
    return thr, idb, pats, overflow


def model_group(y0, thr, idb, pats, mode):
    """(unsafe, g[16]) for the group starting at code phase y0 (half chips)."""
    f = np.float32(y0 - math.floor(y0))
    bi = int(f * np.float32(BINS))
    t = thr[bi]
    unsafe = not (abs(np.float32(f - t)) >= DELTA)
    idn = int(idb[bi]) + (1 if f >= t else 0)
    ev = pats[min(idn, 15)]
    g = np.zeros(16, dtype=np.int64)
    if mode == 2:
        for uu in ev:
            g[uu:] += 1
    else:
        g = np.arange(16, dtype=np.int64)
        for uu in ev:
            g[uu:] -= 1
    return unsafe, g


def brute(y0, s):
    ys = np.empty(17)
    y = y0
    for i in range(17):
        ys[i] = y
        y = y + s
    return ys


@pytest.mark.parametrize("rate,mode", [(2.6e6, 1), (2.1e6, 1), (2.76e6, 1), (25e6, 2), (16e6, 2), (100e6, 2)])
def test_pattern_lookup_reproduces_the_sequential_chip_index(rate, mode):
    rng = np.random.default_rng(int(rate) % 1000 + mode)
    n_unsafe = n_checked = 0
    for ch in range(6):
        s = 2.0 * ((1.023e6 + rng.uniform(-2.3, 2.3)) * (1.0 / rate))
        thr, idb, pats, overflow = build_tables(s, mode)
        assert not overflow and not np.isnan(thr[:BINS]).any()
        T = np.sort(1.0 - np.mod(np.arange(1, 16) * s, 1.0))
        y0s = list(rng.uniform(16.0, 8100.0, 1500))
        for t in T:  # fractions a few float ulps around every threshold: only the margin decides there
            for k in (-40, -9, -3, -1, 0, 1, 3, 9, 40):
                y0s.append(float(rng.integers(16, 8000)) + float(t) + k * 2.0 ** -24)
        for y0 in y0s:
            if y0 - math.floor(y0) >= 1.0 or y0 + 16 * s >= 8184.0:
                continue
            unsafe, g = model_group(y0, thr, idb, pats, mode)
            if unsafe:
                n_unsafe += 1
                continue
            ys = brute(y0, s)
            want = np.floor(ys[:16]).astype(np.int64) - int(math.floor(y0))
            assert np.array_equal(g, want), (rate, s, y0, g, want)
            n_checked += 1
    assert n_checked > 8000 and n_unsafe < 0.08 * (n_checked + n_unsafe)  # (the adversarial points are most of the unsafe ones)


def test_group_advance_is_the_sixteenth_sequential_sum():
    rng = np.random.default_rng(9)
    for _ in range(300):
        s = 2.0 * ((1.023e6 + rng.uniform(-2.3, 2.3)) * (1.0 / 2.6e6))
        m = np.float64(s).view(np.uint64)
        es = int((int(m) >> 52) & 0x7FF) - 1023
        sig = (int(m) & ((1 << 52) - 1)) | (1 << 52)
        ctz = (sig & -sig).bit_length() - 1
        tie_lo = 2.0 ** (es + 1 + ctz)
        for y0 in rng.uniform(1.0, 8100.0, 60):
            ys = brute(y0, s)
            S = (y0 + s) - y0
            y16 = math.fma(S, 16.0, y0) if hasattr(math, "fma") else float(np.float64(S) * 16.0 + np.float64(y0))
            same_binade = math.frexp(y16)[1] == math.frexp(y0)[1]
            in_tie = tie_lo <= y0 < 2.0 * tie_lo or tie_lo <= ys[16] < 2.0 * tie_lo
            if same_binade and not in_tie:
                assert y16 == ys[16], (s, y0)


def test_tie_binade_is_where_the_shortcut_fails():
    # a code step whose significand ends in 1000000 (ctz = 6) with exponent -1: ties in [2^6, 2^7), a binade wide enough to
    # hold whole 16-sample groups (the step is 0.79 half chips)
    s = np.float64(0.78692307692307695)
    m = (int(s.view(np.uint64)) >> 7 << 7) | 64
    s = float(np.uint64(m).view(np.float64))
    bad_in, bad_out, n_in = 0, 0, 0
    rng = np.random.default_rng(3)
    for y0 in rng.uniform(16.0, 1000.0, 6000):
        ys = brute(y0, s)
        if math.frexp(ys[16])[1] != math.frexp(y0)[1]:
            continue
        S = (y0 + s) - y0
        y16 = float(np.float64(S) * 16.0 + np.float64(y0))  # exact: 16 S and the sum are representable here
        inside = 64.0 <= y0 < 128.0
        n_in += inside
        if y16 != ys[16]:
            if inside:
                bad_in += 1
            else:
                bad_out += 1
    assert bad_out == 0 and n_in > 100 and bad_in > n_in // 4  # (about half: the y0 whose last bit is odd)

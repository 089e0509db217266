"""CPU model of k_synth_g's arithmetic (synth_group.hip), in numpy float64 / float32 -- the same IEEE operations the kernel makes --
against the reference's sequential recurrences (src/galileo-sdr.cpp:491-532) stepped sample by sample over 1024-sample chunks:
  * carrier: the group's start phase in closed form, p_g = frac(fma(16 g, |d|, p_c)), the DDA word t = fma(511, p_g, bias)
    advanced by t += 511 |d|: |t - (bias + 511 p_sequential)| stays below 2^-28, and the table index it addresses differs from the
    reference's (int)(511 p) ONLY at samples whose fraction word is below 2^7 -- where the kernel lists the group for k_repair_g;
  * code: y_g = fma(16 g, s, y_c) lies within 2^-30 of the sequential phase, and the half chip of every sample of a group --
    floor(y_g) + floor(u s) + [frac(y_g) >= T_u], the fraction and the thresholds T_u = 1 - frac(u s) rounded to float -- equals the
    reference's (int)(2 x) wherever the fraction keeps 2^-22 away from every threshold and from 0 and 1 (the groups the kernel does
    not list).
No GPU, no product code: this pins the arithmetic argument the default kernel rests on."""
from fractions import Fraction

import numpy as np

BIAS = 1049088.0 + 2.0 ** -26
AMB = 128
GRID = 2.0 ** -32
DELTA = np.float32(2.0 ** -22)


def _fma(a, b, c):
    """Correctly rounded a * b + c per element (Fraction -> float is correctly rounded)."""
    out = np.empty(len(a))
    for i in range(len(a)):
        out[i] = float(Fraction(float(a[i])) * Fraction(float(b[i])) + Fraction(float(c[i])))
    return out


def _table_k(t):
    i = np.floor(t - 1048576.0).astype(np.int64)
    return np.where(i < 512, i - 511, (i - 512) % 511)


def _carrier(pm0, d, n_groups=64):
    """pm0: mirrored phases at the chunk start (either sign), d: |step|.  Returns (#index mismatches at unflagged samples,
    #flagged samples, worst |t - (bias + 511 p)|, #samples)."""
    n = pm0.size
    # the reference's recurrence, :531-532, mirrored (IEEE addition and truncation are sign-symmetric)
    P = np.empty((n_groups * 16, n))
    p = pm0.copy()
    for k in range(n_groups * 16):
        P[k] = p
        q = p + d
        p = q - np.trunc(q)
    c = 511.0 * d
    bad = flagged = 0
    worst = 0.0
    for g in range(0, n_groups, 3):
        g16 = np.full(n, 16.0 * g)
        praw = _fma(g16, d, pm0)
        pg = praw - np.trunc(praw)
        t = _fma(np.full(n, 511.0), pg, np.full(n, BIAS))
        for u in range(16):
            pe = P[16 * g + u]
            k_ref = (511.0 * pe).astype(np.int64)  # (int): truncation towards zero, :509
            lo = np.round((t - np.floor(t)) / GRID).astype(np.int64)
            amb = lo < AMB
            mism = _table_k(t) != np.where(k_ref >= 0, k_ref % 511, k_ref)
            bad += int(np.count_nonzero(mism & ~amb))
            flagged += int(np.count_nonzero(amb))
            err = (t - BIAS) - 511.0 * pe
            err = err - 511.0 * np.round(err / 511.0)
            worst = max(worst, float(np.max(np.abs(err))))
            t = t + c
    return bad, flagged, worst, len(range(0, n_groups, 3)) * 16 * n


def test_carrier_index_differs_only_where_the_group_is_listed():
    rng = np.random.default_rng(20260930)
    bad = flagged = total = 0
    worst = 0.0
    for rep in range(4):
        n = 600
        pm = rng.uniform(0.0, 1.0, n)
        pm[: n // 6] = -rng.uniform(0.0, 1.0, n // 6)  # mirrored phase still negative (after a Doppler sign change)
        d = np.abs(rng.uniform(-3500.0, 3500.0, n)) / 2.6e6
        d[n // 2: n // 2 + 30] = rng.uniform(2.0 ** -40, 1e-9, 30)   # steps far below the grid (down to the host's gate)
        d[-30:] = rng.uniform(20000.0, 38000.0, 30) / 2.6e6          # up to the host's gate (16 x 511 d <= 120)
        b, f, w, t = _carrier(pm, d)
        bad += b
        flagged += f
        total += t
        worst = max(worst, w)
    assert bad == 0
    assert worst < 2.0 ** -28, worst
    assert flagged <= 10 * total * 2.0 ** -25 + 20, (flagged, total)  # 2^-25 per channel-sample expected


def test_carrier_adversarial_phases_next_to_index_boundaries():
    """Chunk-start phases placed so that some sample of the chunk lands within a few ulp of k / 511 (an index boundary): exactly the
    samples the list exists for -- every mismatch must be flagged."""
    rng = np.random.default_rng(7)
    n = 1500
    k = rng.integers(0, 511, n).astype(np.float64)
    pm = np.clip(k / 511.0 + rng.integers(-40, 41, n) * 2.0 ** -53, 0.0, np.nextafter(1.0, 0.0))
    d = np.abs(rng.uniform(-3500.0, 3500.0, n)) / 2.6e6
    bad, flagged, worst, total = _carrier(pm, d, n_groups=4)
    assert bad == 0 and flagged >= n // 2 and worst < 2.0 ** -28, (bad, flagged, worst)


def _code(yc, s, n_groups=64):
    """yc: pre-check code phase (half chips, < 8184) at the chunk start, s: step in half chips.  Returns (#half-chip mismatches in
    unlisted groups, #listed groups, worst |y_g - sequential y|, #groups)."""
    n = yc.size
    # the reference's recurrence in half chips (y = 2 x: doubling commutes with rounding): wrap check before use (:491), add (:528)
    Hs = np.empty((n_groups * 16, n), dtype=np.int64)
    Ys = np.empty((n_groups * 16, n))
    y = yc.copy()
    w = np.zeros(n, dtype=np.int64)
    for k in range(n_groups * 16):
        ge = y >= 8184.0
        y = np.where(ge, y - 8184.0, y)
        w += ge
        Hs[k] = y.astype(np.int64) + 8184 * w  # the half chip the sample reads, unwrapped
        Ys[k] = y + 8184.0 * w
        y = y + s
    us = np.arange(16)[:, None] * s[None, :]
    fl_us = np.floor(us).astype(np.int64)
    T = (1.0 - (us - np.floor(us))).astype(np.float32)  # (u = 0: T = 1, never reached)
    bad = listed = 0
    worst = 0.0
    for g in range(n_groups):
        A = _fma(np.full(n, 16.0 * g), s, yc)
        worst = max(worst, float(np.max(np.abs(A - Ys[16 * g]))))
        H0 = np.floor(A).astype(np.int64)
        f = (A - np.floor(A)).astype(np.float32)
        undec = (f < DELTA) | (f > np.float32(1.0) - DELTA) | (f >= np.float32(1.0))
        undec |= (np.abs(f[None, :] - T[1:]) < DELTA).any(axis=0)
        pred = H0[None, :] + fl_us + (f[None, :] >= T)
        pred[0] = H0
        mism = (pred != Hs[16 * g: 16 * g + 16]).any(axis=0)
        bad += int(np.count_nonzero(mism & ~undec))
        listed += int(np.count_nonzero(undec))
    return bad, listed, worst, n_groups * n


def test_half_chips_differ_only_where_the_group_is_listed():
    rng = np.random.default_rng(20260931)
    n = 300
    yc = rng.uniform(0.0, 8184.0, n)
    yc[:40] = 8184.0 - rng.uniform(0.0, 900.0, 40)       # the chunk's code wrap inside it
    fc = 1.023e6 + rng.uniform(-3500.0, 3500.0, n) * 0.0006493506493506494
    s = 2.0 * (fc * (1.0 / 2.6e6))
    bad, listed, worst, total = _code(yc, s)
    assert bad == 0
    assert worst < 2.0 ** -30, worst
    assert listed <= 10 * total * 32 * 2.0 ** -22 + 5, (listed, total)  # 16 thresholds x 2 x 2^-22 per group expected


def test_half_chips_adversarial_phases_next_to_the_thresholds():
    """Chunk-start phases a few ulp from an integer (sample 0's own boundary) and from the pattern thresholds of some group."""
    rng = np.random.default_rng(11)
    n = 300
    s = np.full(n, 2.0 * ((1.023e6 + 1.3) * (1.0 / 2.6e6)))
    base = rng.integers(0, 8000, n).astype(np.float64)
    u = rng.integers(0, 16, n)
    g = rng.integers(0, 64, n)
    # y_c such that y_c + (16 g + u) s is within a few ulp(8192) of an integer
    yc = base + np.ceil((16 * g + u) * s) - (16 * g + u) * s + rng.integers(-6, 7, n) * 2.0 ** -40
    yc = np.mod(yc, 8184.0)
    bad, listed, worst, total = _code(yc, s)
    assert bad == 0 and listed >= n // 2, (bad, listed)


def _network_masks(f, s):
    """The four stage masks of window form 4 as synth_group.hip builds them with the patterns: sample u reads field u - h(u) of the
    window, h(u) = u - floor(f + u s); read from the output back to the window its value passes p0 = u, p1 = p0 - (h & 1), p2 = p1 -
    (h & 2), p3 = p2 - (h & 4), and stage k shifts the field AT p_k by 2^k fields iff bit k of h is set."""
    m = [0, 0, 0, 0]
    for u in range(16):
        h = u - int(np.floor(f + u * s))
        pp = u
        for k in range(4):
            if h & (1 << k):
                m[k] |= 3 << (2 * pp)
            pp -= h & (1 << k)
    return m


def test_window_form_4_shift_network_is_the_gather():
    """Every pattern of every code step between 4/15 and 0.74 half chips per sample (and beyond: any step below 1): the four-stage
    network -- bfi(M3, x << 16, x), bfi(M2, x << 8, x), bfi(M1, x << 4, x), bfi(M0, x << 2, x) -- puts field floor(f + u s) of the
    window at sample u, for random windows."""
    rng = np.random.default_rng(5)
    bfi = lambda m, a, b: (m & a) | (~m & b & 0xFFFFFFFF)
    n = 0
    for s in np.concatenate([rng.uniform(0.2, 0.9999, 3000), 2.046 / np.array([2.8, 3.0, 3.5, 4.0, 4.092, 5.0, 6.5, 7.0, 7.69])]):
        T = np.sort(1.0 - ((np.arange(1, 16) * s) % 1.0))
        edges = np.concatenate([[0.0], T, [1.0]])
        for lo, hi in zip(edges[:-1], edges[1:]):  # one pattern per interval between two thresholds
            if hi - lo < 1e-9:
                continue
            f = 0.5 * (lo + hi)
            m0, m1, m2, m3 = _network_masks(f, s)
            w = int(rng.integers(0, 1 << 32))
            x = w
            x = bfi(m3, (x << 16) & 0xFFFFFFFF, x)
            x = bfi(m2, (x << 8) & 0xFFFFFFFF, x)
            x = bfi(m1, (x << 4) & 0xFFFFFFFF, x)
            x = bfi(m0, (x << 2) & 0xFFFFFFFF, x)
            for u in range(16):
                g = int(np.floor(f + u * s))
                assert (x >> (2 * u)) & 3 == (w >> (2 * g)) & 3, (s, f, u)
            n += 1
    assert n > 30000

"""Synthetic kernel-boundary workloads: per-epoch channel parameters with the same
distributions the survey specifies, generated on the host with a fixed seed.

  M-SYN12: 12 channels PRN 1..12, f_carr ~ U(-3500, 3500) Hz drifting -0.05 Hz/epoch,
           f_code = 1.023e6 + f_carr * 0.0006493506493506494 (src/gal-sig.cpp:320, include/constants.h:130),
           code_phase0 ~ U[0, 4092), carr_phase0 ~ U[0, 1), ibit0 ~ U{0..499}, page symbols Bernoulli(1/2);
           1199 epochs x 260000 samples (the 120 s / 2.6 MS/s configuration of BASELINE.json).
  M-SYN24: 24 channels, 2.5 M samples/epoch (25 MS/s).
  M-DYN:   M-SYN12 with Doppler from a 10 Hz circular-motion track (r = 100 m, v = 10 m/s).

Generator: numpy's PCG64 (`np.random.default_rng(seed)`), seed 20241008 -- SURVEY.md 8(d) names std::mt19937_64 with the same
seed: the DISTRIBUTIONS are the survey's, the stream is not (a C++ generator has no place in this Python module, and the
checksums the bench reports since round 1 are PCG64's).

The code phase of epoch e+1 continues from epoch e the way the reference's geometry would make it
(code_phase0 advances by samples*f_code*delt modulo 4092, symbol counter accordingly), so page flips
and symbol wraps land mid-epoch exactly as in a real scenario.
"""
import numpy as np

from .synth import CHAN_EPOCH_DTYPE, GAL_CH_RESTART

CODE_FREQ_E1 = 1.023e6
CARR_TO_CODE_E1 = 0.0006493506493506494
SEED = 20241008


def _random_pages(rng, shape):
    w = rng.integers(0, 2**32, size=shape + (16,), dtype=np.uint64).astype(np.uint32)
    w[..., 15] &= (1 << (500 - 480)) - 1  # symbols 500..511 do not exist
    return w


def make_synthetic(n_epochs=1199, n_chan=12, n_slots=16, samples_per_epoch=260000, sample_rate=2.6e6, seed=SEED,
                   drift_hz_per_epoch=-0.05, doppler_span=3500.0, dyn_track=False, prns=None):
    rng = np.random.default_rng(seed)
    p = np.zeros((n_epochs, n_slots), dtype=CHAN_EPOCH_DTYPE)
    if n_chan > n_slots:
        raise ValueError("n_chan > n_slots")
    prns = list(prns) if prns is not None else [(i % 50) + 1 for i in range(n_chan)]
    delt = 1.0 / sample_rate
    e = np.arange(n_epochs, dtype=np.float64)
    for j in range(n_chan):
        f0 = rng.uniform(-doppler_span, doppler_span)
        f_carr = f0 + drift_hz_per_epoch * e
        if dyn_track:
            # line-of-sight velocity of a 10 m/s circular track (r = 100 m) seen at a random azimuth/elevation
            az = rng.uniform(0, 2 * np.pi)
            el = rng.uniform(np.radians(10), np.radians(80))
            omega = 10.0 / 100.0
            t = 0.1 * e
            v_los = 10.0 * np.cos(el) * np.sin(omega * t - az)
            f_carr = f_carr + v_los / 0.1902936727983649
        f_code = CODE_FREQ_E1 + f_carr * CARR_TO_CODE_E1
        cp0 = rng.uniform(0.0, 4092.0)
        ib0 = int(rng.integers(0, 500))
        # continue code phase / symbol counter across epochs as geometry would
        adv = samples_per_epoch * f_code * delt  # chips per epoch
        tot = cp0 + np.concatenate(([0.0], np.cumsum(adv[:-1])))
        wraps = np.floor(tot / 4092.0)
        code_phase0 = tot - wraps * 4092.0
        code_phase0 = np.clip(code_phase0, 0.0, np.nextafter(4092.0, 0.0))
        ibit0 = (ib0 + wraps.astype(np.int64)) % 500
        p["prn"][:, j] = prns[j]
        p["ibit0"][:, j] = ibit0
        p["f_carr"][:, j] = f_carr
        p["f_code"][:, j] = f_code
        p["code_phase0"][:, j] = code_phase0
        p["carr_phase0"][:, j] = 0.0
        p["flags"][:, j] = 0
        p["flags"][0, j] = GAL_CH_RESTART
        p["carr_phase0"][0, j] = rng.uniform(0.0, 1.0)
    pages = _random_pages(rng, (n_epochs, n_chan))
    p["page_next"][:, :n_chan, :] = pages
    p["page_init"][0, :n_chan, :] = _random_pages(rng, (n_chan,))
    return p


def m_syn12(n_epochs=1199):
    return make_synthetic(n_epochs=n_epochs, n_chan=12, n_slots=16, samples_per_epoch=260000, sample_rate=2.6e6,
                          seed=SEED)


def m_syn24(n_epochs=5999):
    return make_synthetic(n_epochs=n_epochs, n_chan=24, n_slots=24, samples_per_epoch=2500000, sample_rate=25e6,
                          seed=SEED + 1)


def m_dyn(n_epochs=2999):
    return make_synthetic(n_epochs=n_epochs, n_chan=12, n_slots=16, samples_per_epoch=260000, sample_rate=2.6e6,
                          seed=SEED + 2, dyn_track=True)

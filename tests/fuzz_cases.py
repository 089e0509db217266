"""Random batch generator of the parity soaks (tests/test_parity_gpu.py and tools/fuzz_parity.py): randomly shaped
batches -- slots, channels, epoch length, sample rate, chunking, Doppler incl. tiny / zero / sign flips / few-bit steps,
channels appearing, vanishing and being re-allocated, symbol counters near the page flip, code phases near the wrap.
Test infrastructure; the sequence of draws is pinned by tests that name a case by its seed."""
import numpy as np

RESTART = 1


def random_case(pkg, rng, big=False, group=False):
    """group=True: batches the default kernel of the reference geometry can take (k_synth_g: rates in the hold form of the resampled
    windows, automatic chunking; zero / sub-2^-40 Doppler steps still send a batch to the exact-replay kernel)."""
    rate = float(rng.choice([2.2e6, 2.4e6, 2.6e6, 2.6e6, 2.6e6, 2.76e6, 3.0e6, 4.0e6, 5.0e6, 6.5e6, 8e6, 16e6, 25e6, 25e6])) if group else float(rng.choice([2.047e6, 2.0465e6, 2.3e6, 2.6e6, 2.6e6, 2.6e6, 2.75e6, 2.78e6, 3.3e6, 4.0e6, 4.092e6, 5.5e6, 7.0e6, 7.7e6, 8e6, 10e6, 12.5e6, 15.4e6, 16e6, 25e6, 25e6, 40e6]))
    n_slots = int(rng.choice([4, 8, 16, 16, 24, 40, 64]))
    if group and rng.random() < 0.7:
        n_slots = int(rng.choice([8, 16, 16, 16]))
    n_chan = int(rng.integers(1, n_slots + 1))
    n_ep = int(rng.integers(1, 7))
    n_samp = int(rng.choice([rng.integers(16, 3000), rng.integers(3000, 70000), int(rate / 10) if rate <= 4.1e6 else 40000]))
    if big:  # reference geometry, many epochs: legs, translation and the stitcher at work
        rate, n_samp = 2.6e6, 260000
        n_slots = 16
        n_chan = int(rng.integers(6, 17))
        n_ep = int(rng.integers(20, 81))
    span = float(rng.choice([5.0, 300.0, 3500.0, 5000.0]))
    p = pkg.workloads.make_synthetic(n_epochs=n_ep, n_chan=n_chan, n_slots=n_slots, samples_per_epoch=n_samp,
                                     sample_rate=rate, seed=int(rng.integers(1 << 30)), doppler_span=span,
                                     drift_hz_per_epoch=float(rng.choice([-0.05, 0.0, 3.0, -40.0])),
                                     prns=[int(x) for x in (rng.permutation(50)[:n_chan] + 1 if n_chan <= 50 else rng.integers(1, 51, n_chan))])
    for j in range(n_chan):
        r = rng.random()
        if r < 0.15:   # exactly zero or tiny Doppler in some epochs
            e = rng.integers(0, n_ep)
            p["f_carr"][e:, j] = rng.choice([1e-5, 1e-7, -3e-5, 0.02] if group and rng.random() < 0.8 else [0.0, 1e-7, -3e-5, 0.02])
        elif r < 0.3:  # sign flip
            e = rng.integers(0, n_ep)
            p["f_carr"][e:, j] = -p["f_carr"][e:, j]
        elif r < 0.45:  # few-bit steps (ties at the wrap)
            k = int(rng.choice([50, 52, 53, 54]))
            d = p["f_carr"][:, j] / rate
            p["f_carr"][:, j] = np.round(d * 2.0 ** k) / 2.0 ** k * rate
        p["f_code"][:, j] = 1.023e6 + p["f_carr"][:, j] * 0.0006493506493506494
        if rng.random() < 0.25:  # code steps with few significant bits: the tie binade of k_synth's group advance moves up
            k = int(rng.integers(1, 16))
            st = (p["f_code"][:, j] * (1.0 / rate)).astype(np.float64)
            m = st.view(np.uint64)
            m = (m >> np.uint64(k) << np.uint64(k)) | np.uint64(1 << k)
            p["f_code"][:, j] = m.view(np.float64) * rate  # (the product may miss the crafted step by an ulp: still few-bit-ish)
        if rng.random() < 0.3:
            p["ibit0"][0, j] = int(rng.choice([498, 499, 0]))
        if rng.random() < 0.3:
            p["code_phase0"][int(rng.integers(0, n_ep)), j] = float(rng.choice([4091.99, 4092.0 + 0.3, 6137.9, 0.0]))
        if rng.random() < 0.15 and n_ep > 1 and p["prn"][-1, j] > 0:  # re-acquired mid-run: fresh carrier and page
            e = int(rng.integers(1, n_ep))
            if p["prn"][e, j] > 0:
                p["flags"][e, j] = RESTART
                p["carr_phase0"][e, j] = rng.uniform(-0.999, 0.999)
                p["page_init"][e, j] = p["page_next"][(e + 1) % n_ep, j]
        if rng.random() < 0.2 and n_ep > 2:  # vanish
            e = int(rng.integers(1, n_ep))
            p[e:, j] = np.zeros((), dtype=p.dtype)
            if rng.random() < 0.5 and e + 1 < n_ep:  # and come back as another PRN with a fresh carrier
                q = pkg.workloads.make_synthetic(n_epochs=n_ep, n_chan=1, n_slots=1, samples_per_epoch=n_samp,
                                                 sample_rate=rate, seed=int(rng.integers(1 << 30)),
                                                 prns=[int(rng.integers(1, 51))])
                p[e + 1:, j] = q[e + 1:, 0]
                p["flags"][e + 1, j] = RESTART
                p["carr_phase0"][e + 1, j] = rng.uniform(-0.999, 0.999)
                p["page_init"][e + 1, j] = q["page_next"][0, 0]
    chunk = int(rng.choice([0, 0, 0, 4 * int(rng.integers(1, 400)), 16 * int(rng.integers(1, 100))]))
    if group:
        chunk = 0
        for j in range(n_chan):  # phases ON an index boundary of the carrier table (511 p an integer) and next to the code's half chips
            if rng.random() < 0.1:
                p["carr_phase0"][0, j] = float(rng.integers(0, 511)) / 511.0 * float(rng.choice([1.0, -1.0]))
            if rng.random() < 0.1:
                p["code_phase0"][int(rng.integers(0, n_ep)), j] = float(rng.integers(0, 8184)) * 0.5 + float(rng.choice([0.0, 1e-12, -1e-12 + 0.5]))
    return p, n_samp, rate, chunk

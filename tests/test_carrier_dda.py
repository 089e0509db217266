"""GPU parity of the opt-in carrier DDA (GAL_CARRIER_DDA=1; synth_kernels.hip: chan_step_rw_cd): k_synth<.., CD = 1> takes the
carrier table index from a fixed-point DDA instead of the exact FP64 phase, flags the waves that meet an index it cannot be
sure of, and the exact-phase kernel synthesises those waves again.  The result must be the oracle's, bit for bit, like the
default body's; gal_synth_stats_t.window_mode carries + 16 when the DDA form ran."""
import numpy as np
import pytest

from oracle_binding import oracle_run
from test_parity_gpu import _compare

pytestmark = pytest.mark.gpu


@pytest.fixture
def dda(monkeypatch):
    monkeypatch.setenv("GAL_CARRIER_DDA", "1")


@pytest.mark.parametrize("n_chan", [1, 5, 9, 12])
def test_dda_small_batches_bit_exact(pkg, dda, n_chan):
    p = pkg.workloads.make_synthetic(n_epochs=4, n_chan=n_chan, n_slots=16, samples_per_epoch=52000, seed=300 + n_chan)
    _, _, stats = _compare(pkg, p, 52000)
    assert stats["window_mode"] == 17
    _, _, stats = _compare(pkg, p, 52000, chunk_samples=1040)
    assert stats["window_mode"] == 17


def test_dda_is_opt_in(pkg):
    p = pkg.workloads.make_synthetic(n_epochs=2, n_chan=4, n_slots=8, samples_per_epoch=52000, seed=310)
    _, _, stats = _compare(pkg, p, 52000)
    assert stats["window_mode"] == 1


def test_dda_negative_tiny_and_zero_doppler(pkg, dda):
    """Both tables (plain / conjugate), steps far below the DDA's grid, a step of exactly zero (t constant) and phases of
    exactly zero (fraction word = the bias: such a wave is flagged and goes through the exact kernel)."""
    p = pkg.workloads.make_synthetic(n_epochs=5, n_chan=8, n_slots=16, samples_per_epoch=52000, seed=311)
    f = np.array([-3400.0, -1000.0, -3.0, -0.02, 0.0, 2.5, 700.0, 3499.0])
    for e in range(5):
        p["f_carr"][e, :8] = f + 0.01 * e * np.sign(f)
        p["f_code"][e, :8] = 1.023e6 + p["f_carr"][e, :8] * 0.0006493506493506494
    p["carr_phase0"][0, :4] = 0.0
    _, _, stats = _compare(pkg, p, 52000)
    assert stats["window_mode"] == 17


def test_dda_doppler_sign_change_between_epochs(pkg, dda):
    """After a sign change the mirrored phase runs NEGATIVE until it crosses zero: the lower half of the DDA table, whose
    entries follow (int)'s truncation towards zero."""
    p = pkg.workloads.make_synthetic(n_epochs=8, n_chan=3, n_slots=16, samples_per_epoch=52000, seed=312)
    for j in range(3):
        f = np.linspace(40.0, -40.0, 8) * (j + 1)
        p["f_carr"][:, j] = f
        p["f_code"][:, j] = 1.023e6 + f * 0.0006493506493506494
    _, _, stats = _compare(pkg, p, 52000)
    assert stats["window_mode"] == 17
    # ... and with steps that change sign every epoch at a few kHz
    p = pkg.workloads.make_synthetic(n_epochs=6, n_chan=6, n_slots=8, samples_per_epoch=52000, seed=313)
    for e in range(6):
        p["f_carr"][e, :6] *= -1.0 if e & 1 else 1.0
        p["f_code"][e, :6] = 1.023e6 + p["f_carr"][e, :6] * 0.0006493506493506494
    _compare(pkg, p, 52000)


def test_dda_gate_on_the_carrier_step(pkg, dda):
    """Sixteen samples may advance the index by at most the table's extension behind a wrap: batches with a larger carrier
    step (here 60 kHz at 2.6 MS/s) stay on the exact-phase body."""
    p = pkg.workloads.make_synthetic(n_epochs=3, n_chan=4, n_slots=8, samples_per_epoch=52000, seed=314)
    p["f_carr"][:, 1] = 60000.0
    _, _, stats = _compare(pkg, p, 52000)
    assert stats["window_mode"] == 1
    p["f_carr"][:, 1] = 30000.0  # 16 x 511 x 30e3 / 2.6e6 = 94 entries: inside
    _, _, stats = _compare(pkg, p, 52000)
    assert stats["window_mode"] == 17


def test_dda_ragged_sizes_and_more_channels_than_one_launch(pkg, dda):
    for n_samp, chunk in [(1000, 0), (2604, 0), (26000, 100), (26000, 252), (4096, 4)]:
        p = pkg.workloads.make_synthetic(n_epochs=3, n_chan=5, n_slots=8, samples_per_epoch=n_samp, seed=n_samp + 1)
        _compare(pkg, p, n_samp, chunk_samples=chunk)
    # 16 channels: the first launch takes the DDA form, the accumulating one the exact-phase body
    p = pkg.workloads.make_synthetic(n_epochs=3, n_chan=16, n_slots=16, samples_per_epoch=52000, seed=315)
    _compare(pkg, p, 52000)


def test_dda_page_flip_code_wraps_and_state_carry(pkg, dda):
    p = pkg.workloads.make_synthetic(n_epochs=6, n_chan=6, n_slots=16, samples_per_epoch=260000, seed=316)
    p["ibit0"][0, :6] = [499, 498, 480, 476, 0, 250]
    for e in range(1, 6):
        p["ibit0"][e, :6] = (p["ibit0"][0, :6] + 25 * e) % 500
    _, st, _ = _compare(pkg, p[:3], 260000)
    q = p[3:].copy()
    q["flags"][0, :6] = 0  # continues from the state the first half returned
    _compare(pkg, q, 260000, state_in=st)


def test_dda_randomised_soak_slice(pkg, dda):
    from fuzz_cases import random_case

    rng = np.random.default_rng(4242)
    seen = 0
    for c in range(60):
        p, n_samp, rate, chunk = random_case(pkg, rng, big=(c % 20 == 19))
        with pkg.SynthEngine(sample_rate=rate, samples_per_epoch=n_samp, n_slots=p.shape[1], device=0,
                             chunk_samples=chunk) as eng:
            iq, st, stats = eng.run_host(p)
            assert eng.walk_counts()[2] == 0
        seen += stats["window_mode"] == 17
        ref_iq, ref_st = oracle_run(p, n_samp, rate)
        assert np.array_equal(iq, ref_iq) and stats["chain_mismatch"] == 0, (c, rate, p.shape, n_samp, chunk)
        act = ref_st["prn"] > 0
        assert np.array_equal(st["carr_phase"][act].view(np.uint64), ref_st["carr_phase"][act].view(np.uint64))
    assert seen >= 5, seen


def test_dda_full_size_equals_the_exact_body(pkg, monkeypatch):
    """M-SYN12 at BASELINE's size (1199 epochs x 260000 samples x 12 SVs = 3.7e9 channel-samples): a batch this long holds
    dozens of samples whose DDA index is uncertain and a few where it is WRONG (profiles/r03u_dda.md) -- the flagged waves
    must come out of the exact kernel: both forms equal word for word, first and last epochs equal to the oracle."""
    import torch

    p = pkg.workloads.m_syn12()
    outs = []
    for on in ("1", "0"):
        monkeypatch.setenv("GAL_CARRIER_DDA", on)
        with pkg.SynthEngine(samples_per_epoch=260000, n_slots=p.shape[1], device=0) as eng:
            eng.plan(p)
            out = torch.empty(eng.output_bytes() // 2, dtype=torch.int16, device="cuda")
            eng.execute(out.data_ptr())
            _, stats = eng.finish()
            assert stats["chain_mismatch"] == 0 and stats["window_mode"] == (17 if on == "1" else 1), stats
        outs.append(out)
    assert torch.equal(outs[0], outs[1])
    ref_iq, _ = oracle_run(p[:2], 260000, 2.6e6)
    assert np.array_equal(outs[0][: ref_iq.size].cpu().numpy(), ref_iq)


def test_dda_epoch_ranges_of_one_plan(pkg, dda):
    """gal_synth_execute_range with the DDA form: the wave flags are those of the LAUNCH (a range of the plan's epochs), and the
    exact-phase pass over the flagged waves writes into the range's own buffer."""
    import torch

    n = 52000
    p = pkg.workloads.make_synthetic(n_epochs=9, n_chan=12, n_slots=16, samples_per_epoch=n, seed=4322)
    p["carr_phase0"][0, :3] = 0.0  # fraction word = the bias: these waves are flagged for certain
    ref_iq, _ = oracle_run(p, n, 2.6e6)
    with pkg.SynthEngine(samples_per_epoch=n, n_slots=16, device=0) as eng:
        eng.plan(p)
        for world in (1, 2, 3):
            parts = []
            for r in range(world):
                e0, ne = pkg.shard.epoch_range(r, world, p.shape[0])
                out = torch.empty(ne * n * 2, dtype=torch.int16, device="cuda")
                eng.execute(out.data_ptr(), e0, ne)
                _, stats = eng.finish()
                assert stats["chain_mismatch"] == 0 and stats["window_mode"] == 17
                parts.append(out.cpu().numpy())
            assert np.array_equal(np.concatenate(parts), ref_iq), world

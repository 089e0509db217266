"""The oracle's sample loop against the REFERENCE'S OWN TEXT compiled here without stand-ins.

oracle/_ref/libref_loop.so is src/galileo-sdr.cpp:481-539 (+ channel_t, include/structures.h:140-162, + the code
expansion, src/gal-sig.cpp:9-233), cut out of the reference tree AT BUILD TIME and compiled with the reference's flags
around a harness that supplies only the locals the fragment names (oracle/Makefile, oracle/ref_loop_harness.cpp).
tools/make_golden_ref_loop.py ran it and committed per-epoch SHA-256s (tests/golden/ref_loop_sha256.npz); the first
tests require oracle/liboracle.so -- the restatement every parity test of the HIP path rests on -- to hash equal on the
same inputs, anywhere.  Where the library itself is present (the build container, and the GPU box through the
snapshot) the last tests compare the two directly, int16 by int16, on random batches."""
import hashlib
import os

import numpy as np
import pytest

from oracle_binding import oracle_codegen, oracle_run
from ref_loop_binding import ref_loop_available, ref_loop_codegen, ref_loop_run

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FX = np.load(os.path.join(G, "ref_loop_sha256.npz"))
N_KB = 26000


def _sha_rows(iq, n_epochs, n):
    return [hashlib.sha256(iq[e * 2 * n:(e + 1) * 2 * n].tobytes()).digest() for e in range(n_epochs)]


def test_oracle_equals_reference_loop_text_on_g1(pkg):
    rows = np.load(os.path.join(G, "g1_params.npz"))["rows"]
    iq, _ = oracle_run(rows, 260000, 2.6e6)
    assert hashlib.md5(iq.tobytes()).hexdigest() == str(FX["g1_md5"])
    got = _sha_rows(iq, rows.shape[0], 260000)
    for e in range(rows.shape[0]):
        assert got[e] == FX["g1_epoch_sha256"][e].tobytes(), e


@pytest.mark.parametrize("k", [0, 1])
def test_oracle_equals_reference_loop_text_on_kernel_boundary_batches(pkg, k):
    rows = FX["kb%d_rows" % k]
    iq, st = oracle_run(rows, N_KB, 2.6e6)
    got = _sha_rows(iq, rows.shape[0], N_KB)
    for e in range(rows.shape[0]):
        assert got[e] == FX["kb%d_epoch_sha256" % k][e].tobytes(), (k, e)
    assert np.array_equal(st["carr_phase"].view(np.uint64), FX["kb%d_carr_end" % k].view(np.uint64))


def test_fixture_batches_are_what_the_tool_generates(pkg):
    """The committed rows are reproducible from the committed generator (no hand-edited inputs)."""
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(G), "..", "tools"))
    from make_golden_ref_loop import kernel_boundary_batch

    for k in (0, 1):
        assert kernel_boundary_batch(pkg, k).tobytes() == FX["kb%d_rows" % k].tobytes()


needs_ref = pytest.mark.skipif(not ref_loop_available(), reason="oracle/_ref/libref_loop.so is built only where "
                               "/root/reference exists (the fixtures above carry its results everywhere else)")


@needs_ref
def test_reference_code_expansion_equals_oracle(pkg):
    """hex_to_binary_converter + sboc + codegen_E1B/E1C (src/gal-sig.cpp:9-233, compiled from the reference's text): all
    50 PRNs, both codes."""
    for prn in range(1, 51):
        for c in (0, 1):
            assert np.array_equal(ref_loop_codegen(prn, c), oracle_codegen(prn, c)), (prn, c)


@needs_ref
def test_reference_loop_text_equals_oracle_on_random_batches(pkg):
    """Random 2.6 MS/s batches of the parity soak's generator (<= 16 slots; the reference's delt is a constant), split in
    two calls with the state carried: identical IQ and identical end state from both libraries."""
    from fuzz_cases import random_case

    rng = np.random.default_rng(4242)
    done = 0
    while done < 12:
        rows, n_samp, rate, _ = random_case(pkg, rng)[:4]
        if rate != 2.6e6 or rows.shape[1] > 16:
            continue
        n_samp = min(n_samp, 30000)
        a, sa = ref_loop_run(rows, n_samp)
        b, sb = oracle_run(rows, n_samp, 2.6e6)
        assert np.array_equal(a, b), done
        assert sa.tobytes() == sb.tobytes()
        if rows.shape[0] > 1:
            cut = rows.shape[0] // 2
            a1, s1 = ref_loop_run(rows[:cut], n_samp)
            a2, s2 = ref_loop_run(rows[cut:], n_samp, s1)
            assert np.array_equal(np.concatenate([a1, a2]), b)
            assert s2.tobytes() == sb.tobytes()
        done += 1

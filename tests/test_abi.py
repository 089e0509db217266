"""The C-ABI library loads on a CPU-only box and exports every symbol include/galsynth.h declares;
record layouts match the C structs; compute entry points fail loudly without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_all_exported(pkg):
    hdr = open(os.path.join(ROOT, "include", "galsynth.h")).read()
    declared = set(re.findall(r"\b(gal_(?:synth|tables)_\w+)\s*\(", hdr))
    declared -= {"gal_synth_t"}
    assert declared, "no declarations parsed"
    from galileo_sdr_sim_amd import synth

    assert declared == set(synth.EXPORTED_SYMBOLS)
    lib = ctypes.CDLL(synth.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), name


def test_struct_layouts_match_header(pkg):
    assert pkg.CHAN_EPOCH_DTYPE.itemsize == 176
    assert pkg.CHAN_STATE_DTYPE.itemsize == 80
    f = pkg.CHAN_EPOCH_DTYPE.fields
    assert [f[k][1] for k in ("prn", "ibit0", "flags", "f_carr", "f_code", "code_phase0", "carr_phase0", "page_next",
                              "page_init")] == [0, 4, 8, 16, 24, 32, 40, 48, 112]
    g = pkg.CHAN_STATE_DTYPE.fields
    assert [g[k][1] for k in ("carr_phase", "page", "prn")] == [0, 8, 72]
    import ctypes

    st = pkg.synth._Stats
    assert ctypes.CDLL(pkg.synth.LIB_PATH).gal_synth_stats_size() == ctypes.sizeof(st)  # (c_int return: 56 fits)
    assert ctypes.sizeof(st) == 56 and st.ms_repair.offset == 48 and st.kernel_family.offset == 40 and st.repaired_groups.offset == 44 and st.ms_walk.offset == 24 and st.window_mode.offset == 32 and st.synth_runs.offset == 36


def test_version_and_tables_without_gpu(pkg):
    lib = pkg.load_library()
    assert b"gfx950" in lib.gal_synth_version()
    t = pkg.tables()
    assert t["e1b"].shape == (50, 128) and t["cos512"].shape == (512,)
    assert t["cos512"][0] == 250 and t["sin512"][0] == 2 and t["cos512"][128] == -2


def test_page_pack_roundtrip(pkg):
    rng = np.random.default_rng(0)
    sym = rng.integers(0, 2, 500)
    w = pkg.pack_page(sym)
    assert w.shape == (16,) and np.array_equal(pkg.unpack_page(w), sym)
    assert w[15] >> 20 == 0


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="GPU present")
def test_no_cpu_fallback(pkg):
    """Without a GPU the engine refuses to exist rather than computing on the host."""
    with pytest.raises(pkg.GalSynthError) as ei:
        pkg.SynthEngine()
    assert ei.value.code == -3  # GAL_E_DEVICE
    assert pkg.device_count() == 0


def test_product_never_touches_the_oracle():
    """Nothing under the package directory, include/ or bench's timed path may reference oracle/."""
    pkg_dir = os.path.join(ROOT, "galileo-sdr-sim_amd")
    for dirpath, _, files in os.walk(pkg_dir):
        for fn in files:
            if fn.endswith((".py", ".cpp", ".hip", ".h", ".inc")) or fn == "Makefile":
                txt = open(os.path.join(dirpath, fn), errors="ignore").read()
                assert "liboracle" not in txt and "oracle_binding" not in txt and "galsyn_oracle" not in txt, fn


def test_headers_compile_as_c_and_link(pkg, tmp_path):
    """A plain C99 translation unit includes both headers, checks the layouts with _Static_assert, links against the two
    libraries and uses the entry points that need no GPU (tests/abi_c/abi_check.c)."""
    import subprocess

    exe = tmp_path / "abi_check"
    pkg_dir = os.path.join(ROOT, "galileo-sdr-sim_amd")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "tests", "abi_c", "abi_check.c"), "-o", str(exe), "-L", pkg_dir, "-lgalsynth",
                        "-lgalscen", "-Wl,-rpath," + pkg_dir], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    nav = os.path.join(ROOT, "tests", "golden", "20feb2022.rnx")
    r = subprocess.run([str(exe), nav], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "gfx950" in r.stdout and "active channels 9" in r.stdout
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.gpu
def test_a_caller_with_an_older_stats_struct_is_not_overrun(pkg):
    """gal_synth_stats_t only grows at its end; the header's gal_synth_finish / gal_synth_run_host are macros over the _n entry points,
    which copy min(the caller's sizeof, the library's): a caller compiled against the 40-byte struct of 0.2 gets 40 bytes."""
    lib = pkg.load_library()
    n = 26000
    p = pkg.workloads.make_synthetic(n_epochs=2, n_chan=3, n_slots=16, samples_per_epoch=n, seed=5)
    with pkg.SynthEngine(samples_per_epoch=n, n_slots=16, device=0) as eng:
        iq = np.empty(2 * n * 2, dtype=np.int16)
        st = np.zeros(16, dtype=pkg.CHAN_STATE_DTYPE)
        buf = (ctypes.c_ubyte * 64)(*([0xAA] * 64))
        pp = np.ascontiguousarray(p, dtype=pkg.CHAN_EPOCH_DTYPE)
        rc = lib.gal_synth_run_host_n(eng._h, pp.ctypes.data, 2, None, iq.ctypes.data, st.ctypes.data,
                                      ctypes.cast(buf, ctypes.POINTER(pkg.synth._Stats)), 40)
        assert rc == 0
    raw = bytes(buf)
    assert raw[40:] == b"\xaa" * 24 and raw[:40] != b"\xaa" * 40
    assert int.from_bytes(raw[8:12], "little") == 2  # n_epochs

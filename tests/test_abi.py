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
    assert ctypes.CDLL(pkg.synth.LIB_PATH).gal_synth_stats_size() == ctypes.sizeof(st)  # (c_int return: 64 fits)
    assert ctypes.sizeof(st) == 64 and st.ms_plan.offset == 56 and st.ms_h2d.offset == 60 and st.exact_records.offset == 52 and st.ms_repair.offset == 48 and st.kernel_family.offset == 40 and st.repaired_groups.offset == 44 and st.ms_walk.offset == 24 and st.window_mode.offset == 32 and st.synth_runs.offset == 36


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


_CHECKER_WORDS = ("liboracle", "oracle_binding", "galsyn_oracle", "gal_oracle_", "ref_loop", "ref_task", "oracle/", "_ref/", "ref_tables_dump",
                  "ref_generate_frame")


def _code_only(path):
    """The text of a source file without its comments (and, for Python, without docstrings): what can be executed."""
    import ast
    import re

    txt = open(path, errors="ignore").read()
    if path.endswith(".py"):
        tree = ast.parse(txt)
        for node in ast.walk(tree):
            if isinstance(node, (ast.Module, ast.FunctionDef, ast.ClassDef, ast.AsyncFunctionDef)) and node.body:
                first = node.body[0]
                if isinstance(first, ast.Expr) and isinstance(getattr(first, "value", None), ast.Constant) and isinstance(first.value.value, str):
                    node.body[0] = ast.Pass()
        return ast.unparse(tree)
    if os.path.basename(path) == "Makefile":
        return re.sub(r"#.*", "", txt)
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return re.sub(r"//[^\n]*", "", txt)


def test_product_never_touches_the_oracle():
    """The checker is test infrastructure: nothing that can EXECUTE under the package directory or in include/ may name the oracle, its
    binding, or anything built from the reference under oracle/_ref (ref_task, ref_loop, ...) -- comments may say how a thing is tested
    --; and in bench.py only the function cpu_baseline may import or call them (the guard used to look for
    liboracle / oracle_binding / galsyn_oracle only)."""
    import ast

    for top in (os.path.join(ROOT, "galileo-sdr-sim_amd"), os.path.join(ROOT, "include")):
        for dirpath, _, files in os.walk(top):
            for fn in files:
                if fn.endswith((".py", ".cpp", ".hip", ".h", ".inc", ".sh")) or fn == "Makefile":
                    code = _code_only(os.path.join(dirpath, fn))
                    for w in _CHECKER_WORDS:
                        assert w not in code, (fn, w)
    # bench.py: every mention in code lies inside cpu_baseline
    src = open(os.path.join(ROOT, "bench.py")).read()
    tree = ast.parse(src)
    allowed = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "cpu_baseline"]
    assert len(allowed) == 1
    lo, hi = allowed[0].lineno, allowed[0].end_lineno
    names = ("oracle_binding", "ref_loop_binding", "oracle_run", "ref_loop_run", "ref_loop_available", "oracle_lib", "oracle_matches_device")
    for node in ast.walk(tree):
        hit = None
        if isinstance(node, ast.ImportFrom) and node.module in names:
            hit = node.module
        elif isinstance(node, ast.Import) and any(a.name in names for a in node.names):
            hit = "import"
        elif isinstance(node, ast.Name) and node.id in names:
            hit = node.id
        elif isinstance(node, ast.Attribute) and node.attr in names:
            hit = node.attr
        elif isinstance(node, ast.Constant) and isinstance(node.value, str) and any(w in node.value for w in ("liboracle", "_ref/lib", "oracle/_ref")) \
                and not (lo <= node.lineno <= hi) and len(node.value) < 200:
            hit = node.value  # (a short string that could be a path; the long ones are documentation in the line's text)
        if hit is not None:
            assert lo <= node.lineno <= hi, "bench.py:%d uses %r outside cpu_baseline" % (node.lineno, hit)
    # ... and __graft_entry__: only smoke() (and build_oracle, which builds and never calls)
    g = ast.parse(open(os.path.join(ROOT, "__graft_entry__.py")).read())
    ok = [(n.lineno, n.end_lineno) for n in g.body if isinstance(n, ast.FunctionDef) and n.name in ("smoke", "build_oracle")]
    for node in ast.walk(g):
        if (isinstance(node, ast.ImportFrom) and node.module in names) or (isinstance(node, ast.Name) and node.id in names):
            assert any(a <= node.lineno <= b for a, b in ok), node.lineno


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
    which copy min(the caller's sizeof, the library's): a caller compiled against the 40-byte struct of 0.2 gets 40 bytes.  The plain
    SYMBOLS gal_synth_finish / gal_synth_run_host -- what a binary built against the 0.2 header calls -- write exactly those 40 bytes
    too (they used to forward the library's own, larger sizeof)."""
    lib = pkg.load_library()
    n = 26000
    p = pkg.workloads.make_synthetic(n_epochs=2, n_chan=3, n_slots=16, samples_per_epoch=n, seed=5)
    with pkg.SynthEngine(samples_per_epoch=n, n_slots=16, device=0) as eng:
        iq = np.empty(2 * n * 2, dtype=np.int16)
        st = np.zeros(16, dtype=pkg.CHAN_STATE_DTYPE)
        pp = np.ascontiguousarray(p, dtype=pkg.CHAN_EPOCH_DTYPE)
        for call in ("sized", "plain"):
            buf = (ctypes.c_ubyte * 72)(*([0xAA] * 72))
            if call == "sized":
                rc = lib.gal_synth_run_host_n(eng._h, pp.ctypes.data, 2, None, iq.ctypes.data, st.ctypes.data,
                                              ctypes.cast(buf, ctypes.POINTER(pkg.synth._Stats)), 40)
            else:
                rc = lib.gal_synth_run_host(eng._h, pp.ctypes.data, 2, None, iq.ctypes.data, st.ctypes.data,
                                            ctypes.cast(buf, ctypes.POINTER(pkg.synth._Stats)))
            assert rc == 0
            raw = bytes(buf)
            assert raw[40:] == b"\xaa" * 32 and raw[:40] != b"\xaa" * 40, call
            assert int.from_bytes(raw[8:12], "little") == 2  # n_epochs
        # the current struct through the sized entry point: all 64 bytes, the plan's host time among them
        buf = (ctypes.c_ubyte * 72)(*([0xAA] * 72))
        rc = lib.gal_synth_run_host_n(eng._h, pp.ctypes.data, 2, None, iq.ctypes.data, st.ctypes.data,
                                      ctypes.cast(buf, ctypes.POINTER(pkg.synth._Stats)), 64)
        assert rc == 0 and bytes(buf)[64:] == b"\xaa" * 8 and bytes(buf)[56:60] != b"\xaa" * 4

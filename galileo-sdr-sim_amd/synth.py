"""ctypes mirror of include/galsynth.h -- the drop-in boundary of the reference's per-sample loop
(reference src/galileo-sdr.cpp:481-539).  Same names, same argument meaning, same error behaviour as
the C-ABI; records travel as numpy structured arrays whose layout IS the C struct layout.

There is no CPU implementation behind this module: if libgalsynth.so is missing, or no gfx950 device is
usable, every entry point raises.
"""
import ctypes
import os

import numpy as np

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG_DIR, "libgalsynth.so")
# the same sources built with -DGAL_TEST_HOOKS (fault injection for the repair-path tests; tests only)
HOOKS_LIB_PATH = os.path.join(PKG_DIR, "libgalsynth_hooks.so")
GAL_CFG_SINGLE_STREAM = 1
GAL_CFG_EXACT_REPLAY = 4  # always the exact-replay kernel (k_synth), also where k_synth_g could run
GAL_CFG_VERIFY_ALL = 8  # accepted and ignored since 0.4: full verification is the default
GAL_CFG_VERIFY_SAMPLED = 16  # k_synth_g batches: re-walk a rotating eighth of the leg positions of both chains per batch (default: every leg)
GAL_CFG_CBOC = 2  # opt-in CBOC(6,1,1/11) sub-carrier (not in the reference; defined by the oracle's CBOC mode)

GAL_CH_RESTART = 1
GAL_PAGE_WORDS = 16
GAL_N_SYM_PAGE = 500

# gal_chan_epoch_t (176 bytes)
CHAN_EPOCH_DTYPE = np.dtype(
    [
        ("prn", "<i4"),
        ("ibit0", "<i4"),
        ("flags", "<u4"),
        ("reserved", "<u4"),
        ("f_carr", "<f8"),
        ("f_code", "<f8"),
        ("code_phase0", "<f8"),
        ("carr_phase0", "<f8"),
        ("page_next", "<u4", (GAL_PAGE_WORDS,)),
        ("page_init", "<u4", (GAL_PAGE_WORDS,)),
    ],
    align=True,
)
assert CHAN_EPOCH_DTYPE.itemsize == 176

# gal_chan_state_t (80 bytes)
CHAN_STATE_DTYPE = np.dtype(
    [("carr_phase", "<f8"), ("page", "<u4", (GAL_PAGE_WORDS,)), ("prn", "<i4"), ("reserved", "<i4")], align=True
)
assert CHAN_STATE_DTYPE.itemsize == 80


class _Cfg(ctypes.Structure):
    _fields_ = [
        ("sample_rate", ctypes.c_double),
        ("samples_per_epoch", ctypes.c_int32),
        ("n_slots", ctypes.c_int32),
        ("device", ctypes.c_int32),
        ("chunk_samples", ctypes.c_int32),
        ("max_walk_passes", ctypes.c_int32),
        ("flags", ctypes.c_uint32),
        ("reserved", ctypes.c_int32 * 2),
    ]


class _Stats(ctypes.Structure):
    _fields_ = [
        ("walk_passes", ctypes.c_int32),
        ("chain_mismatch", ctypes.c_int32),
        ("n_epochs", ctypes.c_int32),
        ("n_active_max", ctypes.c_int32),
        ("chunk_samples", ctypes.c_int32),
        ("chunks_per_epoch", ctypes.c_int32),
        ("ms_walk", ctypes.c_float),
        ("ms_synth", ctypes.c_float),
        ("window_mode", ctypes.c_int32),
        ("synth_runs", ctypes.c_int32),
        ("kernel_family", ctypes.c_int32),
        ("repaired_groups", ctypes.c_int32),
        ("ms_repair", ctypes.c_float),
        ("exact_records", ctypes.c_int32),
        ("ms_plan", ctypes.c_float),
        ("ms_h2d", ctypes.c_float),
    ]


class GalSynthError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("galsynth error %d: %s" % (code, msg))
        self.code = code


# every symbol include/galsynth.h declares (tests check the library exports all of them)
EXPORTED_SYMBOLS = (
    "gal_synth_version",
    "gal_synth_last_error",
    "gal_synth_device_count",
    "gal_synth_create",
    "gal_synth_destroy",
    "gal_synth_set_stream",
    "gal_synth_plan",
    "gal_synth_plan_async",
    "gal_synth_output_bytes",
    "gal_synth_walk_counts",
    "gal_synth_execute",
    "gal_synth_execute_range",
    "gal_synth_finish",
    "gal_synth_finish_n",
    "gal_synth_stats_size",
    "gal_synth_run_host",
    "gal_synth_run_host_n",
    "gal_tables_e1b",
    "gal_tables_e1c",
    "gal_tables_cos512",
    "gal_tables_sin512",
    "gal_tables_cs25",
)

_libs = {}


def load_library(hooks=False):
    """dlopen libgalsynth.so (built in-tree by build.py).  Raises if it is not there: no fallback.
    hooks=True loads the GAL_TEST_HOOKS build instead (tests of the repair paths only)."""
    if hooks in _libs:
        return _libs[hooks]
    path = HOOKS_LIB_PATH if hooks else LIB_PATH
    if not hooks and os.environ.get("GAL_SYNTH_LIB"):
        # A/B experiments only (build_variant.sh (a tool of rounds 3-5: git history)): another build of the same sources, e.g. other register targets;
        # bench.py marks such a line "variant_lib" -- never a result
        path = os.path.abspath(os.environ["GAL_SYNTH_LIB"])
    if not os.path.exists(path):
        raise RuntimeError(
            "%s not found: build it first (python -c 'import __graft_entry__ as g; g.build()'). "
            "The synthesis engine has no CPU fallback." % path
        )
    lib = ctypes.CDLL(path)
    vp, i32 = ctypes.c_void_p, ctypes.c_int32
    lib.gal_synth_version.restype = ctypes.c_char_p
    lib.gal_synth_last_error.restype = ctypes.c_char_p
    lib.gal_synth_device_count.restype = ctypes.c_int
    lib.gal_synth_create.argtypes = [ctypes.POINTER(_Cfg), ctypes.POINTER(vp)]
    lib.gal_synth_destroy.argtypes = [vp]
    lib.gal_synth_set_stream.argtypes = [vp, vp]
    lib.gal_synth_plan.argtypes = [vp, vp, i32, vp]
    lib.gal_synth_plan_async.argtypes = [vp, vp, i32, vp]
    lib.gal_synth_walk_counts.argtypes = [vp, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64),
                                          ctypes.POINTER(ctypes.c_int64)]
    lib.gal_synth_walk_counts.restype = ctypes.c_int
    lib.gal_synth_output_bytes.argtypes = [vp]
    lib.gal_synth_output_bytes.restype = ctypes.c_size_t
    lib.gal_synth_execute.argtypes = [vp, vp]
    lib.gal_synth_execute_range.argtypes = [vp, vp, ctypes.c_int32, ctypes.c_int32]
    lib.gal_synth_execute_range.restype = ctypes.c_int
    lib.gal_synth_finish.argtypes = [vp, vp, ctypes.POINTER(_Stats)]
    lib.gal_synth_run_host.argtypes = [vp, vp, i32, vp, vp, vp, ctypes.POINTER(_Stats)]
    # the sized entry points (what the header's macros call): the library copies min(our sizeof, its own) bytes of statistics
    lib.gal_synth_finish_n.argtypes = [vp, vp, ctypes.POINTER(_Stats), ctypes.c_size_t]
    lib.gal_synth_run_host_n.argtypes = [vp, vp, i32, vp, vp, vp, ctypes.POINTER(_Stats), ctypes.c_size_t]
    lib.gal_synth_stats_size.restype = ctypes.c_size_t
    for name in ("gal_tables_e1b", "gal_tables_e1c", "gal_tables_cos512", "gal_tables_sin512"):
        getattr(lib, name).restype = vp
    lib.gal_tables_cs25.restype = ctypes.c_uint32
    _libs[hooks] = lib
    return lib


def device_count():
    return int(load_library().gal_synth_device_count())


def tables():
    """The signal tables exactly as the engine uses them (numpy copies)."""
    lib = load_library()

    def arr(ptr, ctype, n, shape):
        buf = (ctype * n).from_address(ptr)
        return np.frombuffer(buf, dtype=np.dtype(ctype)).reshape(shape).copy()

    return {
        "e1b": arr(lib.gal_tables_e1b(), ctypes.c_uint32, 50 * 128, (50, 128)),
        "e1c": arr(lib.gal_tables_e1c(), ctypes.c_uint32, 50 * 128, (50, 128)),
        "cos512": arr(lib.gal_tables_cos512(), ctypes.c_int16, 512, (512,)),
        "sin512": arr(lib.gal_tables_sin512(), ctypes.c_int16, 512, (512,)),
        "cs25": int(lib.gal_tables_cs25()),
    }


def pack_page(symbols):
    """500 symbols {0,1} -> 16 little-endian words, bit i of word i>>5 = symbol i."""
    sym = np.asarray(symbols).astype(np.uint8).ravel()
    assert sym.size == GAL_N_SYM_PAGE
    bits = np.zeros(GAL_PAGE_WORDS * 32, dtype=np.uint8)
    bits[:GAL_N_SYM_PAGE] = sym > 0
    return np.packbits(bits, bitorder="little").view("<u4").copy()


def unpack_page(words):
    w = np.ascontiguousarray(np.asarray(words, dtype="<u4"))
    return np.unpackbits(w.view(np.uint8), bitorder="little")[:GAL_N_SYM_PAGE].copy()


class SynthEngine:
    """One handle per GPU / stream (gal_synth_t).  Thread-compatible, not thread-safe."""

    def __init__(self, sample_rate=2.6e6, samples_per_epoch=260000, n_slots=16, device=-1, chunk_samples=0,
                 max_walk_passes=0, flags=0, test_hooks=False):
        self._lib = load_library(hooks=test_hooks)
        self._h = ctypes.c_void_p()
        cfg = _Cfg(float(sample_rate), int(samples_per_epoch), int(n_slots), int(device), int(chunk_samples),
                   int(max_walk_passes), int(flags))
        self._check(self._lib.gal_synth_create(ctypes.byref(cfg), ctypes.byref(self._h)))
        self.sample_rate = float(sample_rate)
        self.samples_per_epoch = int(samples_per_epoch)
        self.n_slots = int(n_slots)
        self.n_epochs = 0
        self._keep = None

    def _check(self, rc):
        if rc != 0:
            raise GalSynthError(rc, self._lib.gal_synth_last_error().decode())

    def close(self):
        if self._h:
            self._lib.gal_synth_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- boundary calls ------------------------------------------------------------------------
    def _params(self, params):
        p = np.ascontiguousarray(params, dtype=CHAN_EPOCH_DTYPE)
        if p.ndim != 2 or p.shape[1] != self.n_slots:
            raise ValueError("params must have shape [n_epochs, n_slots=%d]" % self.n_slots)
        return p

    def _state(self, state_in):
        if state_in is None:
            return None
        s = np.ascontiguousarray(state_in, dtype=CHAN_STATE_DTYPE)
        if s.shape != (self.n_slots,):
            raise ValueError("state_in must have shape [n_slots]")
        return s

    def set_stream(self, hip_stream):
        """hip_stream: integer hipStream_t (e.g. torch.cuda.current_stream().cuda_stream) or None."""
        self._check(self._lib.gal_synth_set_stream(self._h, ctypes.c_void_p(hip_stream or 0)))

    def plan(self, params, state_in=None, wait=True):
        """wait=False: gal_synth_plan_async -- returns once the upload is enqueued; the next execute's walkers wait for it."""
        p = self._params(params)
        s = self._state(state_in)
        fn = self._lib.gal_synth_plan if wait else self._lib.gal_synth_plan_async
        self._check(fn(self._h, p.ctypes.data, p.shape[0], s.ctypes.data if s is not None else None))
        self.n_epochs = p.shape[0]

    def output_bytes(self):
        return int(self._lib.gal_synth_output_bytes(self._h))

    def walk_counts(self):
        """(legs walked, legs translated, fallbacks since create) of the last finish()."""
        a, b, c = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int64(0)
        self._check(self._lib.gal_synth_walk_counts(self._h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        return int(a.value), int(b.value), int(c.value)

    def execute(self, iq_dev_ptr, first_epoch=0, n_epochs=None):
        """iq_dev_ptr: integer device address (e.g. torch tensor .data_ptr()), 16-byte aligned.  With
        first_epoch / n_epochs only that epoch range of the plan is synthesised (into a buffer of that size)."""
        if first_epoch == 0 and n_epochs is None:
            self._check(self._lib.gal_synth_execute(self._h, ctypes.c_void_p(int(iq_dev_ptr))))
        else:
            n = self.n_epochs - first_epoch if n_epochs is None else n_epochs
            self._check(self._lib.gal_synth_execute_range(self._h, ctypes.c_void_p(int(iq_dev_ptr)), int(first_epoch), int(n)))

    def finish(self):
        st = np.zeros(self.n_slots, dtype=CHAN_STATE_DTYPE)
        stats = _Stats()
        self._check(self._lib.gal_synth_finish_n(self._h, st.ctypes.data, ctypes.byref(stats), ctypes.sizeof(_Stats)))
        return st, {k: getattr(stats, k) for k, _ in _Stats._fields_}

    def run_host(self, params, state_in=None):
        """plan + execute + copy to host.  Returns (iq int16 [n_epochs*N*2], state_out, stats)."""
        p = self._params(params)
        s = self._state(state_in)
        iq = np.empty(p.shape[0] * self.samples_per_epoch * 2, dtype=np.int16)
        st = np.zeros(self.n_slots, dtype=CHAN_STATE_DTYPE)
        stats = _Stats()
        self._check(
            self._lib.gal_synth_run_host_n(
                self._h, p.ctypes.data, p.shape[0], s.ctypes.data if s is not None else None, iq.ctypes.data,
                st.ctypes.data, ctypes.byref(stats), ctypes.sizeof(_Stats)
            )
        )
        self.n_epochs = p.shape[0]
        return iq, st, {k: getattr(stats, k) for k, _ in _Stats._fields_}

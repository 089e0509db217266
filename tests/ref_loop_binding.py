"""ctypes binding of oracle/_ref/libref_loop.so -- TEST INFRASTRUCTURE.  The library is the reference's per-sample loop
compiled from the reference's OWN TEXT (src/galileo-sdr.cpp:481-539, cut out at build time by oracle/Makefile; see
oracle/ref_loop_harness.cpp).  It exists only where /root/reference was present at build time (this container; the
built .so travels to the GPU box with the snapshot).  It checks the oracle; nothing else may use it."""
import ctypes
import os

import numpy as np

from oracle_binding import ORACLE_DIR, _dtypes

REF_LOOP_LIB = os.path.join(ORACLE_DIR, "_ref", "libref_loop.so")
_lib = None


def ref_loop_available():
    return os.path.exists(REF_LOOP_LIB)


def ref_loop_lib():
    global _lib
    if _lib is None:
        lib = ctypes.CDLL(REF_LOOP_LIB)
        lib.ref_loop_run.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                     ctypes.c_void_p, ctypes.c_void_p]
        lib.ref_loop_run.restype = ctypes.c_int
        lib.ref_loop_codegen.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        assert lib.ref_loop_sample_rate() == 2600000 and lib.ref_loop_samples_per_epoch() == 260000
        _lib = lib
    return _lib


def ref_loop_run(params, samples_per_epoch, state_in=None):
    """The reference's loop text over [n_epochs, n_slots <= 16] records at the reference's 2.6 MS/s (delt is the
    reference's own constant).  Returns (iq int16, state_out) like oracle_binding.oracle_run."""
    ep_dt, st_dt = _dtypes()
    p = np.ascontiguousarray(params, dtype=ep_dt)
    n_epochs, n_slots = p.shape
    iq = np.zeros(n_epochs * samples_per_epoch * 2, dtype=np.int16)
    st_out = np.zeros(n_slots, dtype=st_dt)
    st_in = None if state_in is None else np.ascontiguousarray(state_in, dtype=st_dt)
    rc = ref_loop_lib().ref_loop_run(p.ctypes.data, n_epochs, n_slots, samples_per_epoch,
                                     st_in.ctypes.data if st_in is not None else None, iq.ctypes.data, st_out.ctypes.data)
    if rc != 0:
        raise RuntimeError("ref_loop rejected the batch (rc=%d)" % rc)
    return iq, st_out


def ref_loop_codegen(prn, e1c):
    ca = np.zeros(8184, dtype=np.int16)
    ref_loop_lib().ref_loop_codegen(int(prn), int(bool(e1c)), ca.ctypes.data)
    return ca

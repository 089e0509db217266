"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this; the product never does."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_LIB = os.path.join(ORACLE_DIR, "liboracle.so")

_lib = None


def oracle_lib():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_LIB):
            subprocess.run(["make", "-C", ORACLE_DIR, "liboracle.so"], check=True, capture_output=True)
        lib = ctypes.CDLL(ORACLE_LIB)
        lib.gal_oracle_run.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                       ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        lib.gal_oracle_run.restype = ctypes.c_int
        lib.gal_oracle_run_cboc.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        lib.gal_oracle_run_cboc.restype = ctypes.c_int
        lib.gal_oracle_cboc_tables.argtypes = [ctypes.c_void_p] * 4
        lib.gal_oracle_tables.argtypes = [ctypes.c_void_p] * 3
        lib.gal_oracle_codegen.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        _lib = lib
    return _lib


def _dtypes():
    import sys
    sys.path.insert(0, ROOT)
    from __graft_entry__ import load_pkg
    pkg = load_pkg()
    return pkg.CHAN_EPOCH_DTYPE, pkg.CHAN_STATE_DTYPE


def oracle_run(params, samples_per_epoch, sample_rate, state_in=None, clock_read=False, cboc=False):
    """CPU restatement of reference src/galileo-sdr.cpp:481-539 over [n_epochs, n_slots] records.
    Returns (iq int16 [n_epochs*N*2], state_out).  cboc=True: the CBOC(6,1,1/11) opt-in mode, which the reference
    does not have (defined by the oracle itself)."""
    ep_dt, st_dt = _dtypes()
    p = np.ascontiguousarray(params, dtype=ep_dt)
    n_epochs, n_slots = p.shape
    iq = np.zeros(n_epochs * samples_per_epoch * 2, dtype=np.int16)
    st_out = np.zeros(n_slots, dtype=st_dt)
    st_in = None if state_in is None else np.ascontiguousarray(state_in, dtype=st_dt)
    if cboc:
        rc = oracle_lib().gal_oracle_run_cboc(p.ctypes.data, n_epochs, n_slots, samples_per_epoch, float(sample_rate),
                                              st_in.ctypes.data if st_in is not None else None, iq.ctypes.data,
                                              st_out.ctypes.data)
    else:
        rc = oracle_lib().gal_oracle_run(p.ctypes.data, n_epochs, n_slots, samples_per_epoch, float(sample_rate),
                                         st_in.ctypes.data if st_in is not None else None, iq.ctypes.data,
                                         st_out.ctypes.data, int(bool(clock_read)))
    if rc != 0:
        raise RuntimeError("oracle rejected the batch (rc=%d)" % rc)
    return iq, st_out


def oracle_matches_device(out_dev, params, samples_per_epoch, sample_rate, piece=200, state_in=None, cboc=False):
    """EVERY epoch of a device output (torch int16 tensor, [n_epochs * N * 2]) against the oracle, int16 by int16: the oracle
    runs the batch in pieces of `piece` epochs with the channel state carried (host memory stays at one piece), each piece is
    compared with the matching slice of the device buffer.  Returns (int16 values that differ, the oracle's end state)."""
    n_epochs = params.shape[0]
    per = samples_per_epoch * 2
    assert out_dev.numel() == n_epochs * per, (out_dev.numel(), n_epochs, per)
    st, bad = state_in, 0
    for a in range(0, n_epochs, piece):
        b = min(a + piece, n_epochs)
        ref, st = oracle_run(params[a:b], samples_per_epoch, sample_rate, state_in=st, cboc=cboc)
        bad += int(np.count_nonzero(out_dev[a * per:b * per].cpu().numpy() != ref))
    return bad, st


def oracle_tables():
    cos = np.zeros(512, dtype=np.int32)
    sin = np.zeros(512, dtype=np.int32)
    cs = np.zeros(25, dtype=np.int8)
    oracle_lib().gal_oracle_tables(cos.ctypes.data, sin.ctypes.data, cs.ctypes.data)
    return cos, sin, cs


def oracle_cboc_tables():
    t = [np.zeros(512, dtype=np.int32) for _ in range(4)]
    oracle_lib().gal_oracle_cboc_tables(*[x.ctypes.data for x in t])
    return t


def oracle_codegen(prn, e1c):
    ca = np.zeros(8184, dtype=np.int16)
    oracle_lib().gal_oracle_codegen(int(prn), int(bool(e1c)), ca.ctypes.data)
    return ca

"""ctypes mirror of include/galscen.h -- the host scenario front-end (RINEX + position + start time ->
per-epoch channel records).  Reference: galileo_task() around its sample loop,
src/galileo-sdr.cpp:202-352, 438-479, 545-564."""
import ctypes
import os

import numpy as np

from .synth import CHAN_EPOCH_DTYPE, GAL_PAGE_WORDS

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG_DIR, "libgalscen.so")

EXPORTED_SYMBOLS = (
    "gal_scen_open",
    "gal_scen_last_error",
    "gal_scen_total_epochs",
    "gal_scen_start_time",
    "gal_scen_next",
    "gal_scen_close",
    "gal_scen_inav_page",
    "gal_scen_inav_raw",
    "gal_scen_crc24q",
    "gal_scen_eph_count",
    "gal_scen_eph_info",
    "gal_scen_eph_gaps",
    "gal_scen_live_rejected",
)


class _Cfg(ctypes.Structure):
    _fields_ = [
        ("nav_file", ctypes.c_char_p),
        ("motion_file", ctypes.c_char_p),
        ("llh", ctypes.c_double * 3),
        ("have_start", ctypes.c_int32),
        ("start", ctypes.c_int32 * 5),
        ("start_sec", ctypes.c_double),
        ("duration_s", ctypes.c_double),
        ("iono_enable", ctypes.c_int32),
        ("n_slots", ctypes.c_int32),
        ("verbose", ctypes.c_int32),
        ("time_overwrite", ctypes.c_int32),
        ("udp_port", ctypes.c_int32),
        ("strict_eph", ctypes.c_int32),
        ("udp_loopback", ctypes.c_int32),
    ]


class GalScenError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("galscen error %d: %s" % (code, msg))
        self.code = code


_lib = None


def load_library():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("%s not found: run __graft_entry__.build()" % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        vp = ctypes.c_void_p
        lib.gal_scen_open.argtypes = [ctypes.POINTER(_Cfg), ctypes.POINTER(vp)]
        lib.gal_scen_last_error.restype = ctypes.c_char_p
        lib.gal_scen_total_epochs.argtypes = [vp]
        lib.gal_scen_start_time.argtypes = [vp, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_double)]
        lib.gal_scen_next.argtypes = [vp, ctypes.c_int32, vp]
        lib.gal_scen_close.argtypes = [vp]
        lib.gal_scen_eph_gaps.argtypes = [vp]
        lib.gal_scen_live_rejected.argtypes = [vp]
        lib.gal_scen_inav_page.argtypes = [vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_double, vp]
        lib.gal_scen_inav_raw.argtypes = [vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_double, vp]
        lib.gal_scen_crc24q.argtypes = [vp, ctypes.c_int32]
        lib.gal_scen_crc24q.restype = ctypes.c_uint32
        lib.gal_scen_eph_count.argtypes = [vp, ctypes.c_int32]
        lib.gal_scen_eph_info.argtypes = [vp, ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(ctypes.c_int32),
                                          ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_double),
                                          ctypes.POINTER(ctypes.c_double)]
        _lib = lib
    return _lib


def crc24q(bits):
    """CRC-24Q of a bit sequence as the page generator computes it (src/inav-msg.cpp:141-167)."""
    b = np.ascontiguousarray(bits, dtype=np.uint8)
    return int(load_library().gal_scen_crc24q(b.ctypes.data, int(b.size)))


def parse_time(text):
    """'YYYY/MM/DD,hh:mm:ss' as the reference's -t (src/main.cpp:259-271)."""
    date, clock = text.split(",")
    y, m, d = (int(v) for v in date.split("/"))
    hh, mm, ss = clock.split(":")
    return [y, m, d, int(hh), int(mm)], float(ss)


class Scenario:
    """Sequential producer of [n_epochs, n_slots] gal_chan_epoch_t rows."""

    def __init__(self, nav_file, llh=(42.3601, -71.0589, 2.0), start=None, duration_s=300.0, iono_enable=True,
                 n_slots=16, motion_file=None, verbose=False, time_overwrite=False, udp_port=0, strict_eph=False,
                 udp_loopback=False):
        self._lib = load_library()
        cfg = _Cfg()
        cfg.nav_file = os.fsencode(nav_file)
        cfg.motion_file = os.fsencode(motion_file) if motion_file else None
        cfg.llh[:] = [float(v) for v in llh]
        if start is not None:
            ymdhm, sec = parse_time(start) if isinstance(start, str) else (list(start[:5]), float(start[5]))
            cfg.have_start = 1
            cfg.start[:] = ymdhm
            cfg.start_sec = sec
        cfg.duration_s = float(duration_s)
        cfg.iono_enable = 1 if iono_enable else 0
        cfg.n_slots = int(n_slots)
        cfg.verbose = 1 if verbose else 0
        # -T (galscen.h: time_overwrite).  True / 1 / "ref": the reference as built with its own flags -- -t without the range check,
        # no record shifted, an empty sky outside the file's span: the reference's bytes for the reference's command line (True == 1:
        # the bool and the int mean the same).  "shift" / 2: TOC and TOE of every record shifted to the start -- what the
        # option sets out to do (CLI: -T ... --shift-toe)
        if isinstance(time_overwrite, str):
            if time_overwrite not in ("ref", "shift"):
                raise ValueError("time_overwrite: False, True, 'ref', 'shift', 1 or 2")
            cfg.time_overwrite = 1 if time_overwrite == "ref" else 2
        else:
            if int(time_overwrite) not in (0, 1, 2):
                raise ValueError("time_overwrite: False, True, 'ref', 'shift', 1 or 2")
            cfg.time_overwrite = int(time_overwrite)
        cfg.udp_port = int(udp_port)
        cfg.strict_eph = 1 if strict_eph else 0
        cfg.udp_loopback = 1 if udp_loopback else 0
        self._keep = cfg
        self._h = ctypes.c_void_p()
        rc = self._lib.gal_scen_open(ctypes.byref(cfg), ctypes.byref(self._h))
        if rc != 0:
            raise GalScenError(rc, self._lib.gal_scen_last_error().decode())
        self.n_slots = int(n_slots)
        self.total_epochs = int(self._lib.gal_scen_total_epochs(self._h))

    def start_time(self):
        w, s = ctypes.c_int32(), ctypes.c_double()
        self._lib.gal_scen_start_time(self._h, ctypes.byref(w), ctypes.byref(s))
        return w.value, s.value

    def next(self, max_epochs):
        rows = np.zeros((max_epochs, self.n_slots), dtype=CHAN_EPOCH_DTYPE)
        n = self._lib.gal_scen_next(self._h, int(max_epochs), rows.ctypes.data)
        if n < 0:
            raise GalScenError(n, self._lib.gal_scen_last_error().decode())
        return rows[:n]

    def all(self):
        return self.next(self.total_epochs)

    @property
    def eph_gaps(self):
        """(satellite, refresh) pairs at which a channel kept a stale ephemeris record (strict_eph off)."""
        return int(self._lib.gal_scen_eph_gaps(self._h))

    @property
    def live_rejected(self):
        return int(self._lib.gal_scen_live_rejected(self._h))

    def inav_page(self, svid, eph_index, week, sec):
        w = np.zeros(GAL_PAGE_WORDS, dtype="<u4")
        rc = self._lib.gal_scen_inav_page(self._h, int(svid), int(eph_index), int(week), float(sec), w.ctypes.data)
        if rc != 0:
            raise GalScenError(rc, self._lib.gal_scen_last_error().decode())
        return w

    def inav_raw(self, svid, eph_index, week, sec):
        """240 page bits before channel coding: even half (114 + 6 tail) then odd half (114 + 6 tail)."""
        b = np.zeros(240, dtype=np.uint8)
        rc = self._lib.gal_scen_inav_raw(self._h, int(svid), int(eph_index), int(week), float(sec), b.ctypes.data)
        if rc != 0:
            raise GalScenError(rc, self._lib.gal_scen_last_error().decode())
        return b

    def ephemerides(self, svid):
        """[(iodnav, toe_week, toe_sec, toc_sec)] of `svid` in file order."""
        out = []
        for k in range(int(self._lib.gal_scen_eph_count(self._h, int(svid)))):
            iod, wk = ctypes.c_int32(), ctypes.c_int32()
            toe, toc = ctypes.c_double(), ctypes.c_double()
            self._lib.gal_scen_eph_info(self._h, int(svid), k, ctypes.byref(iod), ctypes.byref(wk), ctypes.byref(toe),
                                        ctypes.byref(toc))
            out.append((iod.value, wk.value, toe.value, toc.value))
        return out

    def close(self):
        if self._h:
            self._lib.gal_scen_close(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

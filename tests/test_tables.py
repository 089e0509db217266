"""Signal tables: the packed data the engine ships (csrc/e1_tables.inc) against the reference header
itself (oracle/_ref dump of include/constants.h, built only where /root/reference exists) and against
the committed digest of that dump (tests/golden/tables.sha256), plus the oracle's expansions."""
import hashlib
import os

import numpy as np
import pytest

from oracle_binding import oracle_codegen, oracle_tables

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_DUMP = os.path.join(ROOT, "oracle", "_ref", "ref_tables.txt")
GOLDEN = os.path.join(ROOT, "tests", "golden", "tables.sha256")


def _engine_tables_canonical(pkg):
    """Canonical text form shared with tools/make_golden_tables.py: cos, sin, cs25, then 100 code rows of chips."""
    t = pkg.tables()
    lines = ["cos " + " ".join(str(int(v)) for v in t["cos512"]), "sin " + " ".join(str(int(v)) for v in t["sin512"]),
             "cs25 " + " ".join(str((t["cs25"] >> i) & 1) for i in range(25))]
    for prn in range(50):
        for name in ("e1b", "e1c"):
            bits = np.unpackbits(t[name][prn].view(np.uint8), bitorder="little")[:4092]
            lines.append("%s %d %s" % (name, prn + 1, "".join(str(int(b)) for b in bits)))
    return "\n".join(lines) + "\n"


def _ref_tables_canonical():
    out = {}
    codes = []
    for ln in open(REF_DUMP):
        tok = ln.split()
        if tok[0] in ("cos", "sin", "cs25"):
            out[tok[0]] = tok[0] + " " + " ".join(tok[1:])
        elif tok[0] in ("e1b", "e1c"):
            assert int(tok[2]) == 1023
            bits = "".join(format(int(ch, 16), "04b") for ch in tok[3])
            codes.append("%s %s %s" % (tok[0], tok[1], bits))
    return "\n".join([out["cos"], out["sin"], out["cs25"]] + codes) + "\n"


def test_engine_tables_match_committed_digest(pkg):
    digest = hashlib.sha256(_engine_tables_canonical(pkg).encode()).hexdigest()
    assert digest == open(GOLDEN).read().split()[0]


@pytest.mark.skipif(not os.path.exists(REF_DUMP), reason="reference tree absent (oracle/_ref not built)")
def test_engine_tables_match_reference_header(pkg):
    assert _engine_tables_canonical(pkg) == _ref_tables_canonical()
    assert hashlib.sha256(_ref_tables_canonical().encode()).hexdigest() == open(GOLDEN).read().split()[0]


def test_oracle_expansions(pkg):
    t = pkg.tables()
    cos, sin, cs = oracle_tables()
    assert np.array_equal(cos, t["cos512"]) and np.array_equal(sin, t["sin512"])
    assert np.array_equal(cs, [(t["cs25"] >> i) & 1 for i in range(25)])
    for prn, e1c in [(1, 0), (1, 1), (27, 0), (50, 1)]:
        ca = oracle_codegen(prn, e1c)
        bits = np.unpackbits(t["e1c" if e1c else "e1b"][prn - 1].view(np.uint8), bitorder="little")[:4092]
        chip = np.where(bits > 0, -1, 1)
        assert np.array_equal(ca[0::2], -chip) and np.array_equal(ca[1::2], chip)  # BOC(1,1): [-c, +c]


def test_code_properties(pkg):
    """ICD sanity: memory codes are balanced-ish and distinct; E1-B != E1-C."""
    t = pkg.tables()
    ones = np.array([np.unpackbits(t["e1b"][p].view(np.uint8), bitorder="little")[:4092].sum() for p in range(50)])
    assert np.all(np.abs(ones - 2046) < 120)
    assert len({t["e1b"][p].tobytes() for p in range(50)} | {t["e1c"][p].tobytes() for p in range(50)}) == 100

#!/usr/bin/env python3
"""Writes tests/golden/tables.sha256: digest of the canonical text form of the reference's signal tables
(cos/sin LUT, CS25, 50+50 primary codes as chip strings), produced from oracle/_ref/ref_tables.txt, i.e.
from the reference header itself (/root/reference/include/constants.h).  Run in the build container."""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_tables import _ref_tables_canonical  # noqa: E402

digest = hashlib.sha256(_ref_tables_canonical().encode()).hexdigest()
open(os.path.join(ROOT, "tests", "golden", "tables.sha256"), "w").write(digest + "  reference tables (canonical form)\n")
print(digest)

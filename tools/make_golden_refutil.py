#!/usr/bin/env python3
"""tests/golden/ref_generate_frame.json: what the reference's own utils/generate_frame.cpp prints (I/NAV word 2 of the
ephemeris hard-coded in its main(), packed with the reference's encode_int / encode_uint scaling routines), produced by
oracle/_ref/ref_generate_frame -- that file compiled UNMODIFIED where it lies under /root/reference (it includes
nothing but the standard library, so no stand-in is involved; recipe: oracle/Makefile).  The ephemeris values are
data read off that main() (utils/generate_frame.cpp:182-208).  Run in the build container:
    make -C oracle ref && python tools/make_golden_refutil.py"""
import json
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
exe = os.path.join(ROOT, "oracle", "_ref", "ref_generate_frame")
out = subprocess.run([exe], capture_output=True, text=True, check=True).stdout.strip()
fixture = {
    "source": "stdout of oracle/_ref/ref_generate_frame = /root/reference/utils/generate_frame.cpp compiled unmodified (g++ -O1 -std=c++11)",
    "printed": out,  # the four 32-bit words of I/NAV word 2, upper-case hex WITHOUT zero padding, concatenated
    "ephemeris": {  # utils/generate_frame.cpp:182-208
        "af0": -1.000991e-03, "af1": -8.085976e-12, "af2": 0.0, "aop": -6.302550e-01, "bgde5a": 4.656613e-10,
        "bgde5b": 4.656613e-10, "cic": 9.126961e-08, "cis": -5.774200e-08, "crc": 2.680938e+02, "crs": 1.671562e+02,
        "cuc": 7.903203e-06, "cus": 3.773719e-06, "deltan": 2.967624e-09, "ecc": 2.578078e-04, "idot": 4.385897e-10,
        "inc0": 9.805393e-01, "iodnav": 126, "m0": 3.505807e-01, "omg0": -2.759085e+00, "omgdot": -5.827028e-09,
        "sqrta": 5.440616e+03, "toc": 459600, "toe": 459600,
    },
}
path = os.path.join(ROOT, "tests", "golden", "ref_generate_frame.json")
json.dump(fixture, open(path, "w"), indent=1)
print("wrote", path, out)

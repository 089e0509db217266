"""The CLI keeps the reference's option surface and ishort file contract (src/main.cpp:216-326,
src/galileo-sdr.cpp:326-341,536-542)."""
import hashlib
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "galileo-sdr-sim_amd", "galileo-sdr-sim")
G = os.path.join(ROOT, "tests", "golden")
NAV = os.path.join(G, "20feb2022.rnx")
REF = json.load(open(os.path.join(G, "reference_md5.json")))


def test_cli_usage_and_errors(pkg):
    assert os.path.exists(CLI)
    r = subprocess.run([CLI], capture_output=True, text=True)
    assert r.returncode == 1 and "Usage" in r.stdout
    r = subprocess.run([CLI, "-o", "/tmp/x.bin", "-d", "1"], capture_output=True, text=True)
    assert r.returncode == 1 and "ephemeris/nav_msg file is not specified" in r.stdout
    r = subprocess.run([CLI, "-e", "/nonexistent.rnx", "-d", "1"], capture_output=True, text=True)
    assert r.returncode == 1
    r = subprocess.run([CLI, "-e", NAV, "-t", "2019/01/01,00:00:00", "-d", "1"], capture_output=True, text=True)
    assert r.returncode == 1 and "Invalid start time" in r.stderr


@pytest.mark.gpu
def test_cli_g1_file_md5_equals_reference_output(tmp_path):
    out = tmp_path / "g1.ishort"
    # the reference command line, unchanged (README.md:63 style): -U 1 -b 1 are accepted
    r = subprocess.run([CLI, "-e", NAV, "-l", "-6,51,100", "-t", "2022/02/20,12:00:00", "-d", "10", "-U", "1", "-b",
                        "1", "-I", "1", "-o", str(out), "-B", "32"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    data = out.read_bytes()
    assert len(data) == REF["G1"]["bytes"]
    assert hashlib.md5(data).hexdigest() == REF["G1"]["md5"]


@pytest.mark.gpu
def test_cli_stdout_sink():
    r = subprocess.run([CLI, "-e", NAV, "-l", "-6,51,100", "-t", "2022/02/20,12:00:00", "-d", "10", "-I", "1", "-o",
                        "-"], capture_output=True)
    assert r.returncode == 0
    assert hashlib.md5(r.stdout).hexdigest() == REF["G1"]["md5"]

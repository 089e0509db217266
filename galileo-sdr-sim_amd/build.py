"""In-tree build driver: `make` in this directory (hipcc for gfx950, g++ for host-only pieces)."""
import os
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))


def build_all(verbose=False, targets=("all",)):
    """Compile every native piece of the package for gfx950.  Cross-compiles without a GPU."""
    cmd = ["make", "-C", PKG_DIR, "-j%d" % max(2, min(8, os.cpu_count() or 4))] + list(targets)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
        print(res.stderr)
    if res.returncode != 0:
        raise RuntimeError("native build failed: %s" % " ".join(cmd))
    return True

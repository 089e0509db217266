import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _torch_cuda_first():
    """On the GPU box: bring torch's HIP runtime up BEFORE the first engine library is dlopen()ed.  torch ships its
    own ROCm runtime; initialising it after libgalsynth.so AND the GAL_TEST_HOOKS build had both been loaded made
    torch.cuda report 'No HIP GPUs are available' (order-dependent; seen when test_parity_gpu.py ran on its own)."""
    try:
        import torch

        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass
    yield


@pytest.fixture(scope="session")
def pkg():
    from __graft_entry__ import load_pkg

    p = load_pkg()
    lib_missing = not os.path.exists(os.path.join(ROOT, "galileo-sdr-sim_amd", "libgalsynth.so"))
    if lib_missing:
        p.build_all()
    return p


@pytest.fixture(scope="session")
def gpu_engine_factory(pkg):
    """Creating an engine fails loudly when there is no usable gfx950 device."""

    def make(**kw):
        return pkg.SynthEngine(**kw)

    return make

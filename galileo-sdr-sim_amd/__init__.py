"""galileo-sdr-sim_amd -- MI355X-native Galileo E1B/C baseband IQ synthesis engine.

The directory name carries hyphens (it mirrors the reference repo's name), so import it through
``__graft_entry__.load_pkg()`` which registers it as the module ``galileo_sdr_sim_amd``.

Only what the hot path needs lives here:
  csrc/        HIP kernels + the C-ABI of include/galsynth.h (libgalsynth.so)
  synth.py     ctypes mirror of that C-ABI (records as numpy structured arrays)
  scenario.py  ctypes mirror of include/galscen.h (host scenario front-end, libgalscen.so)
  shard.py     per-rank work split + report reductions (torch.distributed: RCCL / gloo)
  workloads.py synthetic kernel-boundary workloads of SURVEY.md §8(d) (M-SYN12, M-SYN24, M-DYN)
  build.py     in-tree build driver (hipcc --offload-arch=gfx950)
"""
from .synth import (  # noqa: F401
    CHAN_EPOCH_DTYPE,
    CHAN_STATE_DTYPE,
    GAL_CH_RESTART,
    GalSynthError,
    SynthEngine,
    device_count,
    load_library,
    pack_page,
    tables,
    unpack_page,
)
from . import workloads  # noqa: F401
from . import scenario  # noqa: F401
from . import shard  # noqa: F401
from .scenario import GalScenError, Scenario  # noqa: F401
from .build import build_all  # noqa: F401

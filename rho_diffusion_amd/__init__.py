"""rho_diffusion_amd: MI355X-native DDPM training + sampling engine behind the class surface of
intel/rho-diffusion (``registry``, ``models.UNet`` = "UNetv2", ``diffusion.DDPM``, schedules).

Host code is Python on PyTorch-ROCm (memory, streams, torch.distributed); all arithmetic of the hot
path runs in hand-written HIP kernels for gfx950 (librho_hip.so, C ABI in include/rho_hip.h)."""
__version__ = "0.1.0"

from .registry import registry  # noqa: F401
from . import layers  # noqa: F401  (registers GroupNorm32 / conv_nd / ...)
from . import models  # noqa: F401  (registers UNetv2, MultiEmbeddings, ...)
from . import diffusion  # noqa: F401  (registers schedules)
from . import config  # noqa: F401,E402
from . import data  # noqa: F401,E402  (registers SphericalHarmonicDataset)


def install_alias(name: str = "rho_diffusion") -> None:
    """Make ``import rho_diffusion`` (and ``from rho_diffusion.diffusion import DDPM``, ``from rho_diffusion.registry import
    registry``, ``from rho_diffusion.config import ExperimentConfig`` ...) resolve to THIS package, so scripts written against the
    reference run unchanged.  Every already-imported submodule is registered under the alias too: a plain
    ``sys.modules["rho_diffusion"] = rho_diffusion_amd`` would let ``import rho_diffusion.registry`` execute registry.py a second
    time and create a second, empty registry."""
    import sys
    me = __name__
    for mod_name, mod in list(sys.modules.items()):
        if mod_name == me or mod_name.startswith(me + "."):
            sys.modules[name + mod_name[len(me):]] = mod
    # module names that differ from the reference's layout
    sys.modules[name + ".diffusion.diffusers"] = sys.modules[me + ".diffusion.diffusers_ddpm"]

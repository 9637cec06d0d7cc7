"""rho_diffusion_amd: MI355X-native DDPM training + sampling engine behind the class surface of
intel/rho-diffusion (``registry``, ``models.UNet`` = "UNetv2", ``diffusion.DDPM``, schedules).

Host code is Python on PyTorch-ROCm (memory, streams, torch.distributed); all arithmetic of the hot
path runs in hand-written HIP kernels for gfx950 (librho_hip.so, C ABI in include/rho_hip.h)."""
__version__ = "0.1.0"

from .registry import registry  # noqa: F401
from . import layers  # noqa: F401  (registers GroupNorm32 / conv_nd / ...)
from . import models  # noqa: F401  (registers UNetv2, MultiEmbeddings, ...)
from . import diffusion  # noqa: F401  (registers schedules)

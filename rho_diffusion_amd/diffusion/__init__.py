from . import schedule  # noqa: F401
from .schedule import LinearSchedule, CosineBetaSchedule, SigmoidSchedule  # noqa: F401
from .abstract_diffusion import AbstractDiffusionPipeline  # noqa: F401
from .ddpm import DDPM  # noqa: F401
from .gaussian_diffusion import GaussianDiffusionPipeline  # noqa: F401
from .diffusers_ddpm import DDPMScheduler, DiffusersDDPMPipeline  # noqa: F401

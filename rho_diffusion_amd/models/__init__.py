from .common import SinusoidalPositionEmbedding, sinosoidal_position_embedding  # noqa: F401
from .conditioning import MultiEmbeddings  # noqa: F401
from .unet_v2 import UNet, ResBlock, AttentionBlock, Upsample, Downsample, TimestepEmbedSequential  # noqa: F401

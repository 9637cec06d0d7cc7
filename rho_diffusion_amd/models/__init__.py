from .common import SinusoidalPositionEmbedding, sinosoidal_position_embedding  # noqa: F401
from .conditioning import MultiEmbeddings  # noqa: F401
from .unet_v2 import UNet, ResBlock, AttentionBlock, Upsample, Downsample, TimestepEmbedSequential  # noqa: F401
from .unet import UNetBlock2d, UNetBlock3d, UNetV1  # noqa: F401  (legacy UNet: registry name "UNet"; ``models.UNet`` here is UNetv2)
UNetv2 = UNet  # noqa: F401  (the reference's export name for the v2 class)

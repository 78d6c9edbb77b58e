"""Drop-in name for the reference's `model` module (reference model.py): `from model import Generator`."""
from cagc.model import *  # noqa: F401,F403
from cagc.model import (Blur, ConstantInput, ConvLayer, Discriminator, Downsample, EqualConv2d, EqualLinear,  # noqa: F401
                        Generator, ModulatedConv2d, NoiseInjection, PixelNorm, ResBlock, ScaledLeakyReLU, StyledConv,
                        ToRGB, Upsample, make_kernel)

"""Public namespace: `from vegans.GAN import WassersteinGANGP, ...` (reference: vegans/GAN.py:1-12)."""
from .models.gan1v1 import (ConditionalLSGAN, ConditionalPix2Pix, ConditionalVanillaGAN, ConditionalWassersteinGAN,
                            ConditionalWassersteinGANGP, LSGAN, VanillaGAN, WassersteinGAN, WassersteinGANGP)
from .models.lrgan import ConditionalLRGAN, LRGAN

__all__ = [
    "VanillaGAN", "WassersteinGAN", "WassersteinGANGP", "LSGAN", "LRGAN",
    "ConditionalVanillaGAN", "ConditionalWassersteinGAN", "ConditionalWassersteinGANGP", "ConditionalLSGAN",
    "ConditionalLRGAN", "ConditionalPix2Pix",
]

"""rna_gan_amd -- MI355X-native WGAN-GP training path of RNA-GAN (hand-written HIP kernels behind a
C ABI; Python host side mirroring the reference's torchgan plugin interface).  See DESIGN.md."""
from .models import DCGANDiscriminator, DCGANGenerator, DCGANUpGenerator, Discriminator, Generator  # noqa: F401
from .betavae import betaVAE  # noqa: F401
from .losses import (WassersteinDiscriminatorLoss, WassersteinDiscriminatorLossVAE,  # noqa: F401
                     WassersteinGeneratorLoss, WassersteinGeneratorLossVAE, WassersteinGradientPenalty,
                     WassersteinGradientPenaltyVAE)
from .trainer import Trainer  # noqa: F401
from .optim import Adam  # noqa: F401

__all__ = ["DCGANGenerator", "DCGANUpGenerator", "DCGANDiscriminator", "Generator", "Discriminator", "betaVAE", "Trainer", "Adam",
           "WassersteinGeneratorLoss", "WassersteinDiscriminatorLoss", "WassersteinGradientPenalty",
           "WassersteinGeneratorLossVAE", "WassersteinDiscriminatorLossVAE", "WassersteinGradientPenaltyVAE"]

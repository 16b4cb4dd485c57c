"""vqk: MI355X-native VQ-VAE / VQ-GAN train-step kernels behind the reference's module surface."""
from . import _native  # noqa: F401

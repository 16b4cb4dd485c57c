"""``BaseVQVAE``: pre/post-processing + the abstract token API of the reference
(vqvae/modules/abstract_modules/base_autoencoder.py:6-93).

The reference normalises with kornia's ``Normalize(0.5, 0.5)`` / ``Denormalize`` and, in training, applies
kornia's RandomResizedCrop + RandomHorizontalFlip.  Here normalisation is the closed form
``clamp(x,0,1)*2-1`` (the train step itself uses the fused HIP ``vqk_preprocess`` kernel instead of this
method).  The random augmentation (``training=True``) is one fused HIP kernel, ``vqk_augment_preprocess``: per-sample
RandomResizedCrop(scale .7-1, ratio 1) + RandomHorizontalFlip with device-side draws (``ops.random_crop_params``);
kornia is not in the reference tree, so its exact random stream is unpinned."""
from abc import ABC, abstractmethod

import torch


class BaseVQVAE(ABC):

    def __init__(self, image_size: int):
        super().__init__()
        self.image_size = image_size
        self.scheduler = None
        self.train_epoch_usage_count = None
        self.val_epoch_usage_count = None

    @torch.no_grad()
    def preprocess_batch(self, images: torch.Tensor, training: bool = False):
        """images (B,C,H,W) in [0,1] -> (-1,1); training=True additionally applies the crop / flip augmentation"""
        if training:
            from ... import ops
            n, c, h, w = images.shape
            box, flip = ops.random_crop_params(n, h, w, images.device)
            x, _ = ops.raw_augment_preprocess(images.float(), box, flip, torch.float32, want_target=False)
            return x[:, :3].contiguous()
        return (torch.clamp(images, 0., 1.) - 0.5) / 0.5

    @torch.no_grad()
    def preprocess_visualization(self, images: torch.Tensor):
        """(-1,1) -> [0,1]"""
        return torch.clip(images * 0.5 + 0.5, 0, 1)

    @abstractmethod
    def get_tokens(self, images: torch.Tensor) -> torch.Tensor:
        """(B,3,H,W) -> (B,S) codebook indices"""

    @abstractmethod
    def quantize(self, images: torch.Tensor) -> torch.Tensor:
        """(B,3,H,W) -> (B,S,D) quantized vectors"""

    @abstractmethod
    def reconstruct(self, images: torch.Tensor) -> torch.Tensor:
        """(B,3,H,W) -> (B,3,H,W) reconstructions"""

    @abstractmethod
    def reconstruct_from_tokens(self, tokens: torch.Tensor) -> torch.Tensor:
        """(B,S) -> (B,3,H,W)"""

"""Quantizer plugin base -- the reference's ``BaseVectorQuantizer`` contract
(vqvae/modules/abstract_modules/base_quantizer.py:6-102): owns the ``nn.Embedding`` codebook
(``state_dict`` key ``codebook.weight``), uniform init, ``codes_to_vec``, usage statistics and dead-code
re-initialisation.  Sub-classes implement ``forward`` / ``vec_to_codes`` on the vqk kernels."""
from abc import ABC, abstractmethod

import torch
import torch.distributed as dist
from torch import nn


class BaseVectorQuantizer(ABC, nn.Module):

    def __init__(self, num_embeddings: int, embedding_dim: int):
        super().__init__()
        self.num_embeddings = num_embeddings
        self.embedding_dim = embedding_dim
        self.codebook = nn.Embedding(num_embeddings, embedding_dim)
        self.kl_warmup = None
        self.temp_decay = None
        self.compute_dtype = torch.float32        # dtype of the quantized output handed to the decoder
        self.last_hist = None                     # int32 [K] code histogram of the latest forward (device)

    def init_codebook(self) -> None:
        nn.init.uniform_(self.codebook.weight, -1 / self.num_embeddings, 1 / self.num_embeddings)

    @abstractmethod
    def forward(self, x: torch.Tensor):
        """x (B,D,H,W) -> (quantized (B,D,H,W), codes (B,H*W) int64 detached, latent loss 0-dim)"""

    @abstractmethod
    def vec_to_codes(self, x: torch.Tensor) -> torch.Tensor:
        """x (B,D,H,W) -> codes (B,H*W) int64"""

    @torch.no_grad()
    def get_codebook(self) -> torch.Tensor:
        return self.codebook.weight

    @torch.no_grad()
    def codes_to_vec(self, codes: torch.Tensor) -> torch.Tensor:
        """codes (B,N) -> (B,N,D)"""
        return self.get_codebook()[codes]

    def get_codebook_usage(self, index_count: torch.Tensor):
        """index_count (K,) -> (usage probabilities (K,), perplexity, percentage of codes used)"""
        p = index_count / torch.sum(index_count)
        perplexity = torch.exp(-torch.sum(p * torch.log(p + 1e-10), dim=-1)).sum().item()
        used = torch.count_nonzero(p).item() * 100 / index_count.shape[0]
        return p, perplexity, used

    @torch.no_grad()
    def reinit_unused_codes(self, codebook_usage: torch.Tensor):
        """codes never used (p == 0) are overwritten with codes sampled in proportion to p (base_quantizer.py:82-102).
        Data parallel: the draw is made on rank 0 and broadcast, so that every replica rewrites the same rows with the same
        codes (the caller all-reduces the usage histogram first) -- replicas never re-synchronise their parameters."""
        unused = torch.nonzero(codebook_usage == 0).squeeze(1)
        if unused.numel() == 0:
            return
        picks = torch.multinomial(codebook_usage.float(), unused.numel(), replacement=True)
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            from ...optim import broadcast_
            broadcast_(picks, src=0)                 # async + wait: no event on a stream that may capture next (optim.all_reduce_sum)
        self.codebook.weight[unused] = self.codebook.weight[picks]

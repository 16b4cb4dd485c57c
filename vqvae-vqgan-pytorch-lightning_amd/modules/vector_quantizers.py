"""Quantizers on the vqk kernels; same classes / ctor signatures / return conventions as the reference's
``vqvae/modules/vector_quantizers.py`` (VectorQuantizer :8-84, EMAVectorQuantizer :87-203,
EntropyVectorQuantizer :277-381, GumbelVectorQuantizer :206-274).

The nearest-codeword search is one exact-fp32 MFMA kernel that never materialises the [N,K] distance
matrix or a one-hot; the reference's association order of the three distance terms is kept so that the
indices are bit-exact.  The EMA statistics are all-reduced over the data-parallel
ranks (a capability the reference lacks -- its ranks silently diverge)."""
import torch
import torch.distributed as dist

from .. import ops
from .abstract_modules.base_quantizer import BaseVectorQuantizer
from .autoencoder import Conv2d


EMA_COLLECTIVES = [0, 0]        # collectives issued / bytes (bench.py reports both per step)


def reduce_ema_stats(stats: torch.Tensor, local_batch: int, force_collective: bool = False) -> float:
    """Collective #2: sum the packed ``[counts(K) | dw(K*D)]`` statistics of every data-parallel rank with
    ONE all-reduce, in place; returns the Laplace-smoothing constant of vector_quantizers.py:164 for the reduced
    statistics = the GLOBAL batch (world * per-rank batch), so that the update equals the reference's single-process EMA
    on the rank-concatenated batch.  Host logic only (no kernel): callable on CPU tensors under gloo."""
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    if world > 1 or (force_collective and dist.is_available() and dist.is_initialized()):
        from ..optim import _track
        work = _track(dist.all_reduce(stats, op=dist.ReduceOp.SUM, async_op=True))     # async + wait: see optim.all_reduce_sum
        if work is not None:
            work.wait()
        EMA_COLLECTIVES[0] += 1
        EMA_COLLECTIVES[1] += stats.numel() * stats.element_size()
    return float(local_batch * world)


def _flat_view(z: torch.Tensor):
    b, d, h, w = z.shape
    return z.permute(0, 2, 3, 1).reshape(b * h * w, d)


class VectorQuantizer(BaseVectorQuantizer):
    def __init__(self, num_embeddings: int, embedding_dim: int, commitment_cost: float = 0.25):
        super().__init__(num_embeddings, embedding_dim)
        self.commitment_cost = commitment_cost

    def forward(self, x: torch.Tensor):
        q, idx, loss, hist = ops.VQLookupFn.apply(x, self.codebook.weight, self.commitment_cost, True, 0,
                                                  self.compute_dtype)
        self.last_hist = hist
        return q, idx, loss

    @torch.no_grad()
    def vec_to_codes(self, x: torch.Tensor) -> torch.Tensor:
        z = ops.nhwc(x.to(torch.float32))
        return ops.vq_assign(_flat_view(z), self.codebook.weight.detach().contiguous(), 0).view(x.shape[0], -1)


class EMAVectorQuantizer(BaseVectorQuantizer):
    def __init__(self, num_embeddings: int, embedding_dim: int, commitment_cost: float = 0.25, decay: float = 0.95,
                 epsilon: float = 1e-5):
        super().__init__(num_embeddings, embedding_dim)
        self.commitment_cost = commitment_cost
        self.codebook.requires_grad_(False)
        self.register_buffer('ema_count', torch.zeros(num_embeddings))
        self.register_buffer('ema_weight', torch.empty(num_embeddings, embedding_dim).uniform_(-1 / num_embeddings,
                                                                                                1 / num_embeddings))
        self.decay = decay
        self.epsilon = epsilon
        # The update only changes what the NEXT step looks up (this step's quantized vectors are already gathered and the
        # backward uses the saved ones), so a trainer may take it out of the forward: with ``defer_update`` the forward
        # leaves this rank's packed statistics in ``pending_stats`` and ``finish_update()`` -- the all-reduce over ranks
        # and the update kernel -- runs after the step's captured graph, next to the gradient all-reduce.
        self.defer_update = False
        self.pending_stats = None
        self._pending_batch = 0

    def forward(self, x: torch.Tensor):
        q, idx, loss, hist = ops.VQLookupFn.apply(x, self.codebook.weight, self.commitment_cost, False, 0,
                                                  self.compute_dtype)
        self.last_hist = hist
        if self.training:
            with torch.no_grad():
                z = ops.nhwc(x.detach().to(torch.float32))
                if self.defer_update:
                    self.pending_stats = ops.ema_stats(_flat_view(z), idx.reshape(-1), self.num_embeddings,
                                                       out=self.pending_stats)
                    self._pending_batch = int(x.shape[0])
                else:
                    stats = ops.ema_stats(_flat_view(z), idx.reshape(-1), self.num_embeddings)
                    batch = reduce_ema_stats(stats, int(x.shape[0]))
                    ops.ema_apply(stats, self.ema_count, self.ema_weight, self.codebook.weight.data, self.decay,
                                  self.epsilon, batch)
        return q, idx, loss

    @torch.no_grad()
    def finish_update(self, force_collective: bool = False) -> None:
        """second half of a deferred update: ONE all-reduce of [counts | dw] over the ranks, then the EMA kernel"""
        if self.pending_stats is None:
            return
        batch = reduce_ema_stats(self.pending_stats, self._pending_batch, force_collective)
        ops.ema_apply(self.pending_stats, self.ema_count, self.ema_weight, self.codebook.weight.data, self.decay,
                      self.epsilon, batch)

    @torch.no_grad()
    def vec_to_codes(self, x: torch.Tensor) -> torch.Tensor:
        z = ops.nhwc(x.to(torch.float32))
        return ops.vq_assign(_flat_view(z), self.codebook.weight.detach().contiguous(), 0).view(x.shape[0], -1)


class EntropyVectorQuantizer(BaseVectorQuantizer):
    def __init__(self, num_embeddings: int, embedding_dim: int, ent_loss_ratio: float = 0.1,
                 ent_temperature: float = 0.01, ent_loss_type: str = 'softmax', commitment_cost: float = 0.25):
        super().__init__(num_embeddings, embedding_dim)
        self.ent_loss_ratio = ent_loss_ratio
        self.ent_temperature = ent_temperature
        self.ent_loss_type = ent_loss_type
        self.commitment_cost = commitment_cost

    def forward(self, x: torch.Tensor):
        q, idx, loss, hist = ops.EntropyVQFn.apply(x, self.codebook.weight, self.commitment_cost, self.ent_loss_ratio,
                                                   self.ent_temperature, self.compute_dtype, self.ent_loss_type)
        self.last_hist = hist
        return q, idx, loss

    @torch.no_grad()
    def vec_to_codes(self, x: torch.Tensor) -> torch.Tensor:
        z = ops.nhwc(x.to(torch.float32))
        return ops.vq_assign(_flat_view(z), self.codebook.weight.detach().contiguous(), 1).view(x.shape[0], -1)


class GumbelVectorQuantizer(BaseVectorQuantizer):
    """Input is the encoder's K-channel logit map (B,K,H,W).  Unlike the other quantizers the indices come back
    shaped (B,H,W) -- the reference's own quirk (vector_quantizers.py:243), kept."""

    def __init__(self, num_embeddings: int, embedding_dim: int, straight_through: bool = False, temp: float = 1.0,
                 kl_cost: float = 5e-4):
        super().__init__(num_embeddings, embedding_dim)
        self.x_to_logits = Conv2d(num_embeddings, num_embeddings, 1, bias=True)
        self.straight_through = straight_through
        self.temp = temp
        self.kl_cost = kl_cost
        self.sched_dev = None          # device [temp, kl_cost]: set by MiniTrainer before a hipGraph capture (replays follow the schedule)

    def enable_device_schedule(self, device) -> torch.Tensor:
        if self.sched_dev is None:
            self.sched_dev = torch.tensor([float(self.temp), float(self.kl_cost)], dtype=torch.float32, device=device)
        return self.sched_dev

    def forward(self, x: torch.Tensor, exp_noise: torch.Tensor = None):
        """``exp_noise`` ~ Exp(1) with the shape of x: injected for parity tests; drawn with torch's RNG otherwise."""
        hard = self.straight_through if self.training else True
        logits = self.x_to_logits(ops.nhwc(x.to(self.compute_dtype)), out_dtype=torch.float32)
        if exp_noise is None:
            exp_noise = torch.empty_like(logits).exponential_()
        q, idx, kl, hist = ops.GumbelVQFn.apply(logits, self.codebook.weight, exp_noise, float(self.temp),
                                                float(self.kl_cost), bool(hard), self.compute_dtype,
                                                self.sched_dev if self.training else None)
        self.last_hist = hist
        return q, idx, kl

    def get_consts(self):
        return self.temp, self.kl_cost

    def set_consts(self, temp: float = None, kl_cost: float = None) -> None:
        if temp is not None:
            self.temp = temp
        if kl_cost is not None:
            self.kl_cost = kl_cost
        if self.sched_dev is not None:                     # two scalar fills on the stream: no host synchronisation
            self.sched_dev[0:1].fill_(float(self.temp))
            self.sched_dev[1:2].fill_(float(self.kl_cost))

    @torch.no_grad()
    def vec_to_codes(self, x: torch.Tensor) -> torch.Tensor:
        """hard Gumbel sample of x itself at tau = 1 (the reference skips x_to_logits here, :272-273)"""
        lg = ops.nhwc(x.to(torch.float32))
        noise = torch.empty_like(lg).exponential_()
        _, idx, _, _ = ops.GumbelVQFn.apply(lg, self.codebook.weight, noise, 1.0, 0.0, True, self.compute_dtype)
        return idx

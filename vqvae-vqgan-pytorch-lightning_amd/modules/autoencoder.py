"""Encoder / decoder of the VQ-VAE on the vqk HIP kernels.

Same classes, constructor signatures, attribute names and ``state_dict`` keys as the reference's
``vqvae/modules/autoencoder.py`` (GroupNorm :7-39, ResBlock :42-77, Downsample :80-91, Upsample :94-106,
Encoder :109-143, Decoder :146-180); the arithmetic is the NHWC kernel set behind ``libvqk.so``:
GroupNorm+SiLU is one fused op, the residual add rides in the conv epilogue, the nearest x2 upsample is
folded into the conv's input addressing, tanh into the last conv's epilogue.

``compute_dtype`` (module attribute, set with :func:`set_compute_dtype`) selects the exact-fp32 MFMA path
(parity mode, default) or bf16 storage / bf16 MFMA with fp32 accumulation (throughput mode).
"""
import math

import torch
from torch import nn

from .. import ops


def resolve_compute_dtype(dtype):
    """``torch.float32`` (exact fp32 products: the reference mode), ``torch.bfloat16`` (throughput mode) or the string ``'bf16x3'``:
    fp32 storage and accumulation with every 3x3-conv product evaluated as three bf16 products on the bf16 matrix pipe
    (``ops.set_conv_products``; parity-grade: ~2^-17 relative per product).  -> (storage dtype, conv products)"""
    if dtype == 'bf16x3':
        return torch.float32, 'bf16x3'
    if dtype in ('fp32', 'float32'):
        return torch.float32, 'fp32'
    if dtype in ('bf16', 'bfloat16'):
        return torch.bfloat16, 'fp32'
    if dtype not in (torch.float32, torch.bfloat16):
        raise ValueError("compute dtype must be torch.float32, torch.bfloat16 or 'bf16x3'")
    return dtype, 'fp32'


def set_compute_dtype(module: nn.Module, dtype) -> nn.Module:
    dtype, products = resolve_compute_dtype(dtype)
    for m in module.modules():
        if hasattr(m, 'compute_dtype'):
            m.compute_dtype = dtype
            m.conv_products = products
    return module


class Conv2d(nn.Module):
    """stride-1 'same' conv (k in {1,3}); weight is logical [O,I,k,k] stored channels_last, i.e. the
    kernels' [O][kh][kw][I] layout.  Initialised exactly like ``torch.nn.Conv2d``."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int, bias: bool = True):
        super().__init__()
        if kernel_size not in (1, 3):
            raise ValueError('only 1x1 and 3x3 convolutions are on the path')
        self.in_channels, self.out_channels, self.kernel_size = in_channels, out_channels, kernel_size
        w = torch.empty(out_channels, in_channels, kernel_size, kernel_size)
        nn.init.kaiming_uniform_(w, a=math.sqrt(5))
        self.weight = nn.Parameter(w.contiguous(memory_format=torch.channels_last))
        if bias:
            bound = 1 / math.sqrt(in_channels * kernel_size * kernel_size)
            self.bias = nn.Parameter(torch.empty(out_channels).uniform_(-bound, bound))
        else:
            self.register_parameter('bias', None)

    def forward(self, x, residual=None, ups: bool = False, act: int = 0, out_dtype=None, next_gn: int = 0):
        return ops.conv2d(x, self.weight, self.bias, residual, ups, act, out_dtype, next_gn)


class GroupNorm(nn.Module):
    def __init__(self, num_groups: int, num_channels: int, eps: float = 1e-6):
        super().__init__()
        if num_channels % num_groups != 0:
            raise ValueError('num_channels must be divisible by num_groups')
        self.num_groups, self.eps = num_groups, eps
        self.weight = nn.Parameter(torch.ones(1, num_channels, 1, 1))
        self.bias = nn.Parameter(torch.zeros(1, num_channels, 1, 1))

    def forward(self, x: torch.Tensor, silu: bool = False) -> torch.Tensor:
        return ops.group_norm_silu(x, self.weight, self.bias, self.num_groups, self.eps, silu)


class ResBlock(nn.Module):
    def __init__(self, in_channels: int, out_channels: int = None):
        super().__init__()
        if out_channels is None or out_channels == in_channels:
            out_channels = in_channels
            self.conv_shortcut = None
        else:
            self.conv_shortcut = Conv2d(in_channels, out_channels, 1, bias=False)
        self.norm1 = GroupNorm(32, in_channels, eps=1e-6)
        self.conv1 = Conv2d(in_channels, out_channels, 3, bias=False)
        self.norm2 = GroupNorm(32, out_channels, eps=1e-6)
        self.conv2 = Conv2d(out_channels, out_channels, 3, bias=False)

    def forward(self, x, pool: bool = False, next_gn: int = 0):
        # one fused autograd node: residual add in conv2's epilogue, skip-gradient add in norm1's backward sweep;
        # pool: the Downsample that follows the block rides in conv2's epilogue as well; next_gn: the GroupNorm that
        # reads the result next has that many groups (its statistics ride in conv2's drain on the large maps)
        return ops.res_block(x, self.norm1.weight, self.norm1.bias, self.conv1.weight, self.norm2.weight,
                             self.norm2.bias, self.conv2.weight,
                             None if self.conv_shortcut is None else self.conv_shortcut.weight,
                             self.norm1.num_groups, self.norm1.eps, pool, next_gn)


class Downsample(nn.Module):
    def __init__(self, kernel_size: int = 2, stride: int = 2, padding: int = 0):
        super().__init__()
        if (kernel_size, stride, padding) != (2, 2, 0):
            raise ValueError('only the 2x2 / stride 2 average pool of the reference is implemented')
        self.kernel_size, self.stride, self.padding = kernel_size, stride, padding

    def forward(self, x):
        return ops.avg_pool2x2(x)


class Upsample(nn.Module):
    def __init__(self, channels: int, scale_factor: float = 2.0, mode: str = 'nearest-exact'):
        super().__init__()
        if scale_factor != 2.0 or mode != 'nearest-exact':
            raise ValueError('only the nearest-exact x2 upsample of the reference is implemented')
        self.scale_factor, self.mode = scale_factor, mode
        self.conv = Conv2d(channels, channels, 3, bias=True)

    def forward(self, x, next_gn: int = 0):
        return self.conv(x, ups=True, next_gn=next_gn)


def _to_internal(x: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """layout/dtype plumbing at the module edge: NHWC storage, compute dtype, channels padded with
    zeros to a whole 16-byte chunk (only the 3-channel image needs it)."""
    e = ops.epc(dtype)
    c = x.shape[1]
    if c % e:
        x = torch.nn.functional.pad(x, (0, 0, 0, 0, 0, e - c % e))
    return x.to(dtype=dtype, memory_format=torch.channels_last)


class Encoder(nn.Module):
    def __init__(self, channels: int, num_res_blocks: int, channel_multipliers: tuple, embedding_dim: int):
        super().__init__()
        self.compute_dtype = torch.float32
        self.conv_in = Conv2d(3, channels, 3, bias=False)
        blocks, ch_in = [], channels
        for mult in channel_multipliers:
            ch_out = channels * mult
            for _ in range(num_res_blocks):
                blocks.append(ResBlock(ch_in, ch_out))
                ch_in = ch_out
            blocks.append(Downsample())
        self.blocks = nn.Sequential(*blocks)
        self.final_residual = nn.Sequential(*[ResBlock(ch_in) for _ in range(num_res_blocks)])
        self.norm = GroupNorm(32, ch_in, eps=1e-6)
        self.conv_out = Conv2d(ch_in, embedding_dim, 1, bias=True)
        self.last_cut = None

    def shallow_split(self, max_fraction: float = 0.12):
        """Index of the Downsample that ends the last resolution level whose cumulative parameter count (from the input) is
        still <= ``max_fraction`` of the encoder's, or None.  The high-resolution levels hold few parameters but most of the
        backward's time: cutting the backward there lets the deep levels' gradients be all-reduced under it, and leaves only
        this small head exposed (trainer.MiniTrainer._split_step)."""
        total = sum(p.numel() for p in self.parameters())
        cum, best = sum(p.numel() for p in self.conv_in.parameters()), None
        for i, m in enumerate(self.blocks):
            cum += sum(p.numel() for p in m.parameters())
            if isinstance(m, Downsample) and cum <= max_fraction * total and i + 1 < len(self.blocks):
                best = i
        return best

    def shallow_parameters(self, split):
        mods = list(self.blocks)
        return list(self.conv_in.parameters()) + [p for m in mods[:split + 1] for p in m.parameters()]

    def forward(self, x, cut_after=None):
        """``cut_after``: index into ``self.blocks`` (``shallow_split()``): the autograd graph is cut behind that module and
        the pair (tensor before the cut, detached leaf after it) is left in ``self.last_cut``"""
        ops.set_conv_products(getattr(self, 'conv_products', 'fp32'))
        x = _to_internal(x, self.compute_dtype)
        mods = list(self.blocks)
        i = 0
        # every ResBlock / pooled output is read next by a 32-group GroupNorm (the next block's norm1, finally self.norm)
        gn = self.norm.num_groups
        # ... and so is conv_in's output when a ResBlock follows (its norm1): the sums ride in the first conv's store loop
        x = self.conv_in(x, next_gn=gn if (mods and isinstance(mods[0], ResBlock)) else 0)
        self.last_cut = None
        while i < len(mods):                                  # ResBlock + Downsample pairs run as one fused op
            step = 2 if (isinstance(mods[i], ResBlock) and i + 1 < len(mods) and isinstance(mods[i + 1], Downsample)) else 1
            cut_here = cut_after is not None and i <= cut_after < i + step
            ngn = 0 if cut_here else gn                       # (the fused GroupNorm sums are keyed to the producing tensor: not across a cut)
            if step == 2:
                x = mods[i](x, pool=True, next_gn=ngn)
            elif isinstance(mods[i], ResBlock):
                x = mods[i](x, next_gn=ngn)
            else:
                x = mods[i](x)
            i += step
            if cut_here and torch.is_grad_enabled() and x.requires_grad:
                xc = x.detach().requires_grad_(True)
                self.last_cut = (x, xc)
                x = xc
        for blk in self.final_residual:
            x = blk(x, next_gn=gn)
        x = self.norm(x, silu=True)
        return self.conv_out(x, out_dtype=torch.float32)      # the quantizer always sees fp32 latents


class Decoder(nn.Module):
    def __init__(self, channels: int, num_res_blocks: int, channel_multipliers: tuple, embedding_dim: int):
        super().__init__()
        self.compute_dtype = torch.float32
        ch_in = channels * channel_multipliers[-1]
        self.conv_in = Conv2d(embedding_dim, ch_in, 3, bias=True)
        self.initial_residual = nn.Sequential(*[ResBlock(ch_in) for _ in range(num_res_blocks)])
        blocks = []
        for i in reversed(range(len(channel_multipliers))):
            ch_out = channels * channel_multipliers[i - 1] if i > 0 else channels
            for _ in range(num_res_blocks):
                blocks.append(ResBlock(ch_in, ch_out))
                ch_in = ch_out
            blocks.append(Upsample(ch_out))
        self.blocks = nn.Sequential(*blocks)
        self.norm = GroupNorm(32, channels, eps=1e-6)
        self.conv_out = Conv2d(channels, 3, 3, bias=True)

    def forward_padded(self, x):
        """reconstruction with its zero pad channels ([N, 4 or 8, H, W], NHWC) -- what the fused loss reads"""
        ops.set_conv_products(getattr(self, 'conv_products', 'fp32'))
        x = _to_internal(x, self.compute_dtype)
        x = self.conv_in(x)
        # a ResBlock / Upsample output is read next by a 32-group GroupNorm (the next block's norm1, finally self.norm)
        # unless an Upsample conv follows it
        gn = self.norm.num_groups
        mods = list(self.initial_residual) + list(self.blocks)
        for i, blk in enumerate(mods):
            to_conv = i + 1 < len(mods) and isinstance(mods[i + 1], Upsample)
            x = blk(x, next_gn=0 if to_conv else gn)
        x = self.norm(x, silu=True)
        return self.conv_out(x, act=1)                        # tanh in the epilogue

    def forward(self, x):
        return self.forward_padded(x)[:, :3]

"""Seed-regenerated inputs / weights and compact summaries for the FULL-SIZE fixtures (tests/golden/full_*.npz).

At BASELINE.json's sizes the weights (42 M parameters) and most outputs are too large to commit, so both sides -- the
generator that imports the reference (``make_golden_full.py``, build container only) and the tests (CPU oracle pin,
``-m gpu`` parity) -- regenerate the SAME inputs and weights from seeds with the helpers below, and the fixture stores
what the reference produced from them: integer outputs exactly (indices), small tensors in full, large tensors as a
slice plus a ``summary`` (sum, l2 norm, four projections onto seeded random directions).  A projection of an error
vector e onto a unit-variance random direction has standard deviation ||e||, so ``check_summary`` bounds ||e|| / ||t||.

No reference source here: pure torch, usable on the GPU box.
"""
import zlib

import numpy as np
import torch

N_PROJ = 4


def gen_for(name: str, seed: int) -> torch.Generator:
    return torch.Generator().manual_seed((zlib.crc32(name.encode()) + 7919 * seed) & 0x7FFFFFFF)


def fill_named(named, seed: int, kind: str = 'autoencoder') -> None:
    """In-place, order-independent fill of parameters by NAME (the reference's and this build's modules share
    ``state_dict`` keys but not constructor RNG order).

    autoencoder: conv weights U(-b, b) with b = 1/sqrt(fan_in) (torch's default conv init bound), GroupNorm weight
    1 + 0.1 N(0,1), biases 0.1 N(0,1) (conv biases and GroupNorm shifts non-trivial), codebook untouched.
    discriminator: weights N(0,1) (the equalised-lr convention, discriminator.py:103,152), biases 0.1 N(0,1)."""
    with torch.no_grad():
        for name, p in named:
            g = gen_for(name, seed)
            parts = name.split('.')
            is_norm = len(parts) >= 2 and parts[-2].startswith('norm')
            if name.endswith('codebook.weight'):
                continue
            if name.endswith('bias'):
                v = torch.randn(p.shape, generator=g) * 0.1
            elif is_norm:
                v = 1.0 + 0.1 * torch.randn(p.shape, generator=g)
            elif kind == 'discriminator':
                v = torch.randn(p.shape, generator=g)
            else:
                fan_in = int(np.prod(p.shape[1:]))
                b = 1.0 / fan_in ** 0.5
                v = (torch.rand(p.shape, generator=g) * 2 - 1) * b
            p.copy_(v.to(p.dtype))


def summary(t: torch.Tensor, name: str) -> np.ndarray:
    """[sum, l2 norm, proj_0..3] in float64; projections onto N(0,1) directions seeded by ``name``"""
    x = t.detach().to('cpu', torch.float64).contiguous().reshape(-1)
    out = [x.sum().item(), x.norm().item()]
    g = gen_for('proj.' + name, 0)
    for _ in range(N_PROJ):
        r = torch.randn(x.numel(), generator=g, dtype=torch.float64)
        out.append(torch.dot(x, r).item())
    return np.asarray(out, dtype=np.float64)


def check_summary(t: torch.Tensor, ref: np.ndarray, name: str, tol: float) -> float:
    """assert the logical (NCHW-order) tensor ``t`` matches the reference summary to relative l2 error ~``tol``;
    returns the worst projection error relative to the reference norm"""
    got = summary(t, name)
    norm = max(float(ref[1]), 1e-30)
    worst = float(np.abs(got[2:] - ref[2:]).max() / norm)
    assert worst <= 3.0 * tol, (name, 'projection error / norm', worst, 'tol', tol)
    assert abs(got[1] - ref[1]) <= 3.0 * tol * norm, (name, 'norm', got[1], ref[1])
    return worst


# ------------------------------------------------------------------------------------------------------------------
# inputs of the full-size cases (shared by make_golden_full.py and the tests)
# ------------------------------------------------------------------------------------------------------------------
AE_FULL = dict(channels=128, num_res_blocks=2, channel_multipliers=(1, 2, 2, 4))


def ema_full_inputs():
    """config 3: EMA quantizer K=1024, D=256, bs=32 at 256x256 -> latents [32,256,16,16], N=8192; trained-like state"""
    g = torch.Generator().manual_seed(3003)
    z = torch.randn(32, 256, 16, 16, generator=g) * 0.5
    fz = z.permute(0, 2, 3, 1).reshape(-1, 256)
    e = fz[torch.randperm(8192, generator=g)[:1024]] + 0.02 * torch.randn(1024, 256, generator=g)
    count = torch.rand(1024, generator=g) * 8.0 + 0.05
    weight = e * count[:, None] + 0.01 * torch.randn(1024, 256, generator=g)
    dq = torch.randn(32, 256, 16, 16, generator=g)
    return dict(z=z, e=e.contiguous(), ema_count=count, ema_weight=weight, dq=dq, beta=0.25, decay=0.95, eps=1e-5)


def entropy_full_inputs(temperature: float, n_img: int = 16):
    """config 5: entropy quantizer K=8192, D=256; n_img=16 -> N=4096 (oracle-feasible), 64 -> N=16384 (BASELINE size)"""
    g = torch.Generator().manual_seed(5005 + n_img)
    z = torch.randn(n_img, 256, 16, 16, generator=g) * 0.3
    fz = z.permute(0, 2, 3, 1).reshape(-1, 256)
    pick = torch.randint(0, fz.shape[0], (8192,), generator=g)
    e = fz[pick] * 0.9 + 0.08 * torch.randn(8192, 256, generator=g)
    dq = torch.randn(n_img, 256, 16, 16, generator=g) * 1e-3
    return dict(z=z, e=e.contiguous(), dq=dq, beta=0.25, ratio=0.1, temperature=temperature)


def gumbel_full_inputs():
    """config 4 quantizer: K=1024, D=256, bs=16 -> encoder logits [16,1024,16,16] (N=4096), injected Exp(1) noise"""
    g = torch.Generator().manual_seed(4004)
    x = torch.randn(16, 1024, 16, 16, generator=g)
    e = (torch.rand(1024, 256, generator=g) * 2 - 1) / 1024            # base_quantizer.py:31
    w = (torch.rand(1024, 1024, 1, 1, generator=g) * 2 - 1) / 32.0     # conv default bound 1/sqrt(1024)
    b = (torch.rand(1024, generator=g) * 2 - 1) / 32.0
    noise = torch.empty(16, 1024, 16, 16).exponential_(generator=g)
    dq = torch.randn(16, 256, 16, 16, generator=g)
    return dict(x=x, e=e, w=w, b=b, noise=noise, dq=dq, tau=1.0, kl_cost=0.00859375)


def disc256_inputs():
    g = torch.Generator().manual_seed(6006)
    x = torch.randn(4, 3, 256, 256, generator=g) * 0.5
    r = torch.randn(4, 1, generator=g)
    return dict(x=x, r=r, seed=6006)


def config1_inputs():
    """config 1: standard_vqvae.yaml at 64x64, bs=8 (latents 4x4, N=128), full architecture, K=1024, D=256"""
    g = torch.Generator().manual_seed(1234)
    images = torch.rand(8, 3, 64, 64, generator=g)
    e = torch.randn(1024, 256, generator=gen_for('quantizer.codebook.weight', 1001)) * 0.05
    return dict(images=images, codebook=e, seed=1001)


def fill_vqgan(m, seed: int) -> None:
    """seeded weights of a whole VQ-GAN ``VQVAE`` (reference or this build: same parameter names): auto-encoder and
    Gumbel logits conv, codebook N(0, 0.3^2), discriminator, the VGG16 backbone of LPIPS and its five lin layers"""
    fill_named([(n, p) for n, p in m.named_parameters() if n.startswith(('encoder.', 'decoder.', 'quantizer.'))], seed)
    fill_named(list(m.criterion.discriminator.named_parameters()), seed + 1, 'discriminator')
    with torch.no_grad():
        cb = m.quantizer.codebook.weight
        cb.copy_(torch.randn(cb.shape, generator=gen_for('quantizer.codebook.weight', seed)) * 0.3)
        for n, p in m.criterion.perceptual_loss.net.named_parameters():     # 'layers.{i}.weight|bias'
            fill_named([(n, p)], seed + 2)
            if n.endswith('bias'):
                p.mul_(0.1)
        for i, lin in enumerate(m.criterion.perceptual_loss.lin):
            w = lin[1].weight
            w.copy_(torch.rand(w.shape, generator=gen_for(f'lin.{i}', seed + 2)))

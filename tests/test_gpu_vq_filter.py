"""The bf16 candidate filter + exact fp32 re-rank form of the nearest-codeword assignment (csrc/vq_filter.hip; reference:
vqvae/modules/vector_quantizers.py:37-44 / :337-343 followed by torch.argmin) must return EXACTLY the indices of the
exact-fp32 MFMA kernel it replaces and of the C oracle -- on well-separated data, on near-ties, on exact ties, on a
collapsed codebook (candidate-list overflow -> in-kernel exact fallback), with K = 8192 (tiles beyond the LDS cache are
recomputed) and with ragged N."""
import importlib

import numpy as np
import pytest
import torch

from oracle import vq_c

pytestmark = pytest.mark.gpu

ops = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd.ops')
native = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd._native')
DEV = 'cuda:0'


def _both(z, e, assoc):
    zd, ed = z.to(DEV).contiguous(), e.to(DEV).contiguous()
    assert ops.VQ_FILTER
    calls = {'n': 0}
    lib = native.lib()
    real = lib.vqk_vq_assign_filtered_f32

    class Counting:
        def __getattr__(self, name):
            if name == 'vqk_vq_assign_filtered_f32':
                def f(*a):
                    calls['n'] += 1
                    return real(*a)
                return f
            return getattr(lib, name)
    saved = native.lib
    native.lib = lambda: Counting()
    try:
        got = ops.vq_assign(zd, ed, assoc).cpu().numpy()
    finally:
        native.lib = saved
    assert calls['n'] == 1                                           # the filter path really ran
    ops.VQ_FILTER = False
    try:
        want = ops.vq_assign(zd, ed, assoc).cpu().numpy()
    finally:
        ops.VQ_FILTER = True
    return got, want


def _trained_like(n, k, g, noise=0.01):
    z = torch.randn(n, 256, generator=g) * 0.36
    e = z[torch.randperm(n, generator=g)[:k]] + noise * torch.randn(k, 256, generator=g)
    return z, e


@pytest.mark.parametrize('assoc', [0, 1])
@pytest.mark.parametrize('case', ['random', 'trained', 'near_ties', 'exact_ties', 'collapsed', 'scaled_small', 'ragged', 'k8192', 'mixed_norms'])
def test_filter_indices_equal_exact_kernel_and_oracle(case, assoc):
    g = torch.Generator().manual_seed(sum(map(ord, case)) + assoc)
    n, k = 8192, 1024
    if case == 'random':                         # uniform tiny codebook against N(0,1) latents: distances bunch up
        z = torch.randn(n, 256, generator=g)
        e = (torch.rand(k, 256, generator=g) * 2 - 1) / k
    elif case == 'trained':
        z, e = _trained_like(n, k, g)
    elif case == 'near_ties':                    # codes differing in the last bits: far inside the bf16 filter's margin
        z, e = _trained_like(n, k, g)
        e[1::2] = e[0::2] * (1 + 2.0 ** -20)
    elif case == 'exact_ties':                   # duplicated codes: the lower index must win
        z, e = _trained_like(n, k, g)
        e[k // 2:] = e[:k // 2]
    elif case == 'collapsed':                    # every code (nearly) the same vector: thousands of candidates per block
        z = torch.randn(n, 256, generator=g)
        e = torch.randn(1, 256, generator=g).repeat(k, 1) + 1e-6 * torch.randn(k, 256, generator=g)
    elif case == 'scaled_small':
        z, e = _trained_like(n, k, g)
        z, e = z * 1e-3, e * 1e-3
    elif case == 'ragged':
        n = 1000 + 17
        z, e = _trained_like(4096, k, g)
        z = z[:n].contiguous()
    elif case == 'k8192':
        n, k = 4096, 8192
        z = torch.randn(n, 256, generator=g) * 0.36
        e = torch.randn(k, 256, generator=g) * 0.36
    else:                                        # code norms spread over four decades
        z = torch.randn(n, 256, generator=g)
        e = torch.randn(k, 256, generator=g) * torch.logspace(-2, 2, k).unsqueeze(1)
    got, want = _both(z, e, assoc)
    assert np.array_equal(got, want), int((got != want).sum())
    rows = slice(0, 512)                          # the scalar C oracle is ~0.5 ms per (row, 1024 codes)
    ref, _, _, _ = vq_c.assign(z[rows].numpy(), e.numpy(), assoc)
    assert np.array_equal(got[rows], ref)
    if case == 'exact_ties':
        assert (got < k // 2).all()

"""GPU parity tests: every HIP kernel, called through the C-ABI (libvqk.so), against the CPU oracle and the
golden vectors captured from the reference.  Run with ``pytest -m gpu`` on an MI355X."""
import importlib

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import vqvae_oracle as O
from oracle import vq_c

pytestmark = pytest.mark.gpu

pkg = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd')
ops = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd.ops')
ae = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd.modules.autoencoder')
vqm = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd.modules.vector_quantizers')

DEV = 'cuda:0'
T = torch.from_numpy


def dev(a, dtype=None):
    t = T(np.ascontiguousarray(a)) if isinstance(a, np.ndarray) else a
    t = t.to(DEV)
    return t.to(dtype) if dtype is not None else t


def close(a, b, rtol=1e-4, atol=1e-5):
    a = a.detach().float().cpu().numpy() if isinstance(a, torch.Tensor) else a
    b = b.detach().float().cpu().numpy() if isinstance(b, torch.Tensor) else b
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


def rel_err(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


# ------------------------------------------------------------------------------------------ VQ
@pytest.mark.parametrize('n,k,d,scale', [(128, 64, 16, 1.0), (1000, 100, 24, 0.36), (33, 1024, 256, 1.0),
                                         (8192, 1024, 256, 1.0), (4096, 1024, 256, 0.01)])
@pytest.mark.parametrize('assoc', [0, 1])
def test_vq_assign_bit_exact_vs_c_oracle(n, k, d, scale, assoc):
    g = torch.Generator().manual_seed(n * 7 + k)
    z = torch.randn(n, d, generator=g) * scale
    e = (torch.rand(k, d, generator=g) * 2 - 1) / k
    if n >= 4096:                              # trained-like codebook + duplicate rows (exact ties)
        e = z[torch.randperm(n, generator=g)[:k]] + 0.01 * torch.randn(k, d, generator=g)
        e[k // 2] = e[3]
    idx = ops.vq_assign(dev(z), dev(e), assoc).cpu().numpy()
    rows = slice(0, min(n, 2048))              # the scalar C oracle is ~0.5 ms per (row, 1024 codes)
    ref, _, _, _ = vq_c.assign(z[rows].numpy(), e.numpy(), assoc)
    assert np.array_equal(idx[rows], ref)
    tref = torch.argmin(O.distances_std(z, e) if assoc == 0 else O.distances_entropy(z, e), dim=1).numpy()
    assert (idx != tref).mean() < 2e-3         # torch's own GEMM order may flip genuine near-ties only
    if n >= 4096:
        assert not (idx == k // 2).any()       # duplicate of row 3: lower index wins


@pytest.mark.parametrize('tag,assoc', [('n1', 0), ('n036', 0), ('ent', 1)])
def test_vq_assign_golden_large(golden, tag, assoc):
    """BASELINE shape, all 8192 rows, against the indices the reference itself produced."""
    g = golden('vq_large')
    gen = torch.Generator().manual_seed(int(g[f'{tag}.seed']))
    z = torch.randn(32, 256, 16, 16, generator=gen) * float(g[f'{tag}.scale'])
    e = (torch.rand(1024, 256, generator=gen) * 2 - 1) / 1024
    fz = z.permute(0, 2, 3, 1).reshape(-1, 256)
    if tag != 'n1':
        e = fz[torch.randperm(8192, generator=gen)[:1024]] + 0.01 * torch.randn(1024, 256, generator=gen)
    idx = ops.vq_assign(dev(fz.contiguous()), dev(e), assoc).cpu().numpy()
    assert np.array_equal(idx, g[f'{tag}.idx'].astype(np.int64).reshape(-1))


@pytest.mark.parametrize('tag', ['std_a', 'std_b', 'std_c'])
def test_vq_standard_module_golden(golden, tag):
    g = golden('vq')
    q = vqm.VectorQuantizer(64, 16, 0.25).to(DEV)
    with torch.no_grad():
        q.codebook.weight.copy_(dev(g[f'{tag}.e']))
    z = dev(g[f'{tag}.z']).requires_grad_(True)
    qz, idx, loss = q(z)
    assert np.array_equal(idx.cpu().numpy(), g[f'{tag}.idx'])
    assert np.array_equal(q.vec_to_codes(z.detach()).cpu().numpy(), g[f'{tag}.codes'])
    close(qz, g[f'{tag}.q'], rtol=1e-5, atol=1e-6)  # reference returns z + (q - z): one extra rounding
    close(loss, g[f'{tag}.loss'], rtol=1e-5, atol=1e-8)
    dz, de = torch.autograd.grad([qz, loss], [z, q.codebook.weight], [dev(g[f'{tag}.dq']), torch.ones((), device=DEV)])
    close(dz, g[f'{tag}.dz'], rtol=1e-5, atol=1e-7)
    close(de, g[f'{tag}.de'], rtol=1e-4, atol=1e-7)


def test_vq_ema_module_trajectory(golden):
    g = golden('vq')
    q = vqm.EMAVectorQuantizer(32, 16, 0.25, 0.95, 1e-5).to(DEV)
    with torch.no_grad():
        q.codebook.weight.copy_(dev(g['ema.e0']))
        q.ema_weight.copy_(dev(g['ema.w0']))
        q.ema_count.copy_(dev(g['ema.c0']))
    q.train()
    for s in range(3):
        z = dev(g[f'ema.z{s}']).requires_grad_(True)
        qz, idx, loss = q(z)
        assert np.array_equal(idx.cpu().numpy(), g[f'ema.idx{s}'])
        close(qz, g[f'ema.q{s}'], rtol=1e-5, atol=1e-6)
        close(loss, g[f'ema.loss{s}'], rtol=1e-5)
        close(q.ema_count, g[f'ema.count{s}'], rtol=1e-5, atol=1e-7)
        close(q.ema_weight, g[f'ema.weight{s}'], rtol=1e-5, atol=1e-7)
        close(q.codebook.weight, g[f'ema.cb{s}'], rtol=1e-5, atol=1e-7)
        dz, = torch.autograd.grad([qz, loss], [z], [dev(g[f'ema.dq{s}']), torch.ones((), device=DEV)])
        close(dz, g[f'ema.dz{s}'], rtol=1e-5, atol=1e-7)


def test_vq_entropy_module_golden(golden):
    g = golden('vq')
    q = vqm.EntropyVectorQuantizer(64, 16, 0.1, 0.01, 'softmax', 0.25).to(DEV)
    with torch.no_grad():
        q.codebook.weight.copy_(dev(g['ent.e']))
    z = dev(g['ent.z']).requires_grad_(True)
    qz, idx, loss = q(z)
    assert np.array_equal(idx.cpu().numpy(), g['ent.idx'])
    assert np.array_equal(q.vec_to_codes(z.detach()).cpu().numpy(), g['ent.idx'])
    close(qz, g['ent.q'], rtol=1e-5, atol=1e-6)
    close(loss, g['ent.loss'], rtol=2e-5, atol=1e-7)
    dz, de = torch.autograd.grad([qz, loss], [z, q.codebook.weight], [dev(g['ent.dq']), torch.ones((), device=DEV)])
    assert rel_err(dz, T(g['ent.dz'])) < 2e-4
    assert rel_err(de, T(g['ent.de'])) < 2e-4


@pytest.mark.parametrize('tag,k,d', [('a', 64, 16), ('b', 128, 32)])
def test_vq_entropy_argmax_module_golden(golden, tag, k, d):
    """ent_loss_type='argmax': one-hot targets with the softmax gradient (vector_quantizers.py:311-315)"""
    g = golden('vq_entropy_argmax')
    q = vqm.EntropyVectorQuantizer(k, d, 0.1, float(g[f'{tag}.temp']), 'argmax', 0.25).to(DEV)
    with torch.no_grad():
        q.codebook.weight.copy_(dev(g[f'{tag}.e']))
    z = dev(g[f'{tag}.z']).requires_grad_(True)
    qz, idx, loss = q(z)
    assert np.array_equal(idx.cpu().numpy(), g[f'{tag}.idx'])
    close(qz, g[f'{tag}.q'], rtol=1e-5, atol=1e-6)
    close(loss, g[f'{tag}.loss'], rtol=2e-5, atol=1e-7)
    dz, de = torch.autograd.grad([qz, loss], [z, q.codebook.weight], [dev(g[f'{tag}.dq']), torch.ones((), device=DEV)])
    assert rel_err(dz, T(g[f'{tag}.dz'])) < 2e-4
    assert rel_err(de, T(g[f'{tag}.de'])) < 2e-4
    with pytest.raises(ValueError):
        vqm.EntropyVectorQuantizer(k, d, 0.1, 0.01, 'hardmax', 0.25).to(DEV)(z)


def test_vq_gumbel_module_golden(golden):
    g = golden('vq')
    q = vqm.GumbelVectorQuantizer(32, 8, False, 0.7, 5e-4).to(DEV)
    with torch.no_grad():
        q.codebook.weight.copy_(dev(g['gum.e']))
        q.x_to_logits.weight.copy_(dev(g['gum.w']))
        q.x_to_logits.bias.copy_(dev(g['gum.b']))
    q.train()
    x = dev(g['gum.x']).requires_grad_(True)
    qz, idx, loss = q(x, exp_noise=dev(g['gum.noise']))
    assert np.array_equal(idx.cpu().numpy(), g['gum.idx'])
    close(qz, g['gum.q'], rtol=1e-4, atol=1e-6)
    close(loss, g['gum.loss'], rtol=1e-4, atol=1e-9)
    gr = torch.autograd.grad([qz, loss], [x, q.codebook.weight, q.x_to_logits.weight, q.x_to_logits.bias],
                             [dev(g['gum.dq']), torch.ones((), device=DEV)])
    for a, key in zip(gr, ('gum.dx', 'gum.de', 'gum.dw', 'gum.db')):
        assert rel_err(a, T(g[key])) < 2e-4, key
    # hard (straight-through) forward picks exactly one code per position
    q.eval()
    qh, idxh, _ = q(x.detach(), exp_noise=dev(g['gum.noise']))
    close(qh, q.codebook.weight.detach()[idxh].permute(0, 3, 1, 2), rtol=1e-6, atol=1e-7)


def test_vq_entropy_large_vs_oracle():
    """K = 1024, N = 2048, D = 256 (config-5-like aspect, bounded for the CPU oracle): loss and both gradients"""
    g = torch.Generator().manual_seed(77)
    z = (torch.randn(8, 256, 16, 16, generator=g) * 0.05)
    e = (torch.randn(1024, 256, generator=g) * 0.05)
    zr, er = z.clone().requires_grad_(True), e.clone().requires_grad_(True)
    qr, idxr, lossr = O.vq_entropy(zr, er, 0.25, 0.1, 0.01)
    dq = torch.randn(qr.shape, generator=g) * 1e-3
    gz, ge = torch.autograd.grad([qr, lossr], [zr, er], [dq, torch.tensor(1.0)])
    q = vqm.EntropyVectorQuantizer(1024, 256, 0.1, 0.01, 'softmax', 0.25).to(DEV)
    with torch.no_grad():
        q.codebook.weight.copy_(dev(e))
    zd = dev(z).requires_grad_(True)
    qz, idx, loss = q(zd)
    assert (idx.cpu() != idxr).float().mean() < 2e-3
    close(loss, lossr, rtol=1e-4, atol=1e-6)
    dz, de = torch.autograd.grad([qz, loss], [zd, q.codebook.weight], [dev(dq), torch.ones((), device=DEV)])
    assert rel_err(dz, gz) < 1e-3 and rel_err(de, ge) < 1e-3


# ------------------------------------------------------------------------------------------ conv
CONV_CASES = [
    # n, cin, cout, h, w, k, ups, bias, residual
    (2, 32, 64, 6, 10, 3, False, False, False),
    (1, 64, 32, 9, 7, 3, False, True, True),
    (2, 128, 128, 16, 16, 3, False, False, True),
    (2, 32, 48, 5, 3, 3, True, True, False),
    (3, 64, 128, 8, 8, 1, False, False, False),
    (2, 16, 32, 4, 4, 3, False, True, False),
    (1, 256, 512, 8, 8, 3, False, False, False),
    (2, 8, 32, 12, 12, 3, False, False, False),
    # halo-kernel eligible (W % 32 == 0, H % 8 == 0 or 16x16 patches; Cin a whole 128-byte chunk)
    (2, 64, 128, 8, 32, 3, False, True, True),
    (1, 128, 160, 16, 64, 3, False, False, False),
    (3, 64, 64, 16, 16, 3, False, False, True),
    (1, 128, 256, 32, 16, 3, False, True, False),
    (2, 64, 64, 8, 16, 3, True, True, False),
    (2, 16, 24, 8, 8, 3, False, True, False),           # partial co / ci tiles in the all-taps wgrad kernel
    (1, 72, 40, 16, 8, 3, False, False, False),
    # one-chunk inputs / outputs on W % 32 == 0 maps: the dedicated edge-conv kernels (bf16)
    (2, 8, 128, 8, 32, 3, False, True, False),
    (1, 8, 64, 6, 64, 3, False, False, False),
    (1, 8, 96, 5, 32, 3, False, True, False),
    (2, 128, 8, 8, 32, 3, False, True, False),
    (1, 64, 8, 16, 64, 3, False, False, False),
    # nearest-x2 upsample convs whose dgrad pools in the epilogue (bf16, Cin % 128 == 0), both patch shapes
    (2, 128, 128, 8, 16, 3, True, True, False),
    (1, 128, 256, 8, 8, 3, True, False, False),
]


def _conv_ref(x, w, b, res, k, ups):
    if ups:
        x = x.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)
    y = F.conv2d(x, w, b, padding=k // 2)
    return y + res if res is not None else y


@pytest.mark.parametrize('case', CONV_CASES)
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_conv_fwd_bwd(case, dtype):
    n, cin, cout, h, w, k, ups, has_b, has_r = case
    g = torch.Generator().manual_seed(hash(case) % 10000)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    b = torch.randn(cout, generator=g) if has_b else None
    s = 2 if ups else 1
    res = torch.randn(n, cout, h * s, w * s, generator=g) if has_r else None
    dy = torch.randn(n, cout, h * s, w * s, generator=g)
    if dtype == torch.bfloat16:                # compare on bf16-representable operands
        x, wt, dy = x.bfloat16().float(), wt.bfloat16().float(), dy.bfloat16().float()
        res = res.bfloat16().float() if res is not None else None
    leaves = [t.clone().requires_grad_(True) for t in (x, wt)] + ([b.clone().requires_grad_(True)] if has_b else []) \
        + ([res.clone().requires_grad_(True)] if has_r else [])
    yr = _conv_ref(leaves[0], leaves[1], leaves[2] if has_b else None, leaves[-1] if has_r else None, k, ups)
    gr = torch.autograd.grad(yr, leaves, dy)

    xd = dev(x, dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wd = dev(wt).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    bd = dev(b).requires_grad_(True) if has_b else None
    rd = dev(res, dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True) if has_r else None
    y = ops.conv2d(xd, wd, bd, rd, ups, 0, None)
    assert y.shape == yr.shape and y.dtype == dtype
    gl = [xd, wd] + ([bd] if has_b else []) + ([rd] if has_r else [])
    gd = torch.autograd.grad(y, gl, dev(dy, dtype).contiguous(memory_format=torch.channels_last))
    tol = 2e-5 if dtype == torch.float32 else 6e-3      # bf16: output rounding only (operands exact)
    assert rel_err(y, yr) < tol
    for a, r in zip(gd, gr):
        assert a.shape == r.shape
        assert rel_err(a, r) < (tol if dtype == torch.float32 else 1e-2)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_conv_kernel_variants_agree(dtype):
    """im2col kernel, halo kernel with LDS-staged weights, halo kernel with register weights: same answer"""
    native = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd._native')
    g = torch.Generator().manual_seed(11)
    x = dev(torch.randn(2, 128, 16, 32, generator=g), dtype).contiguous(memory_format=torch.channels_last)
    w = dev(torch.randn(160, 3, 3, 128, generator=g) * 0.03).reshape(-1)
    outs = []
    try:
        for variant in (0, 2, 1, 4):          # 4: register-weight stream kernel with 256-pixel tiles only (no half tiles)
            native.lib().vqk_conv_set_variant(variant)
            layout = ops.weight_layout(dtype, 2, 16, 32, 128, 160, 3, False)
            assert layout == (1 if variant in (1, 4) else 0)
            wq = ops.pack_weights(w, dtype, 160, 128, 3, False, layout)
            outs.append(ops.raw_conv_fprop(x, wq, None, None, 3, False, 0, dtype, 160, layout))
    finally:
        native.lib().vqk_conv_set_variant(-1)
    assert rel_err(outs[1], outs[0]) < 1e-3 and rel_err(outs[2], outs[0]) < 1e-3 and rel_err(outs[3], outs[0]) < 1e-3


@pytest.mark.parametrize('h,w', [(8, 32), (16, 16), (16, 64)])
@pytest.mark.parametrize('has_b,has_r', [(False, False), (True, True), (False, True)])
def test_conv_pooled_epilogue_matches_conv_then_pool(h, w, has_b, has_r):
    """fused 2x2 pooling in the stream kernel's epilogue == conv kernel followed by the pool kernel (bf16)"""
    g = torch.Generator().manual_seed(h * 100 + w + has_b)
    dt = torch.bfloat16
    x = dev(torch.randn(2, 64, h, w, generator=g), dt).contiguous(memory_format=torch.channels_last)
    wt = dev(torch.randn(128, 3, 3, 64, generator=g) * 0.05).reshape(-1)
    b = dev(torch.randn(128, generator=g)) if has_b else None
    r = dev(torch.randn(2, 128, h, w, generator=g), dt).contiguous(memory_format=torch.channels_last) if has_r else None
    layout = ops.weight_layout(dt, 2, h, w, 64, 128, 3, False)
    assert layout == 1 and ops.can_pool_epilogue(dt, 128, layout)
    wq = ops.pack_weights(wt, dt, 128, 64, 3, False, layout)
    for scale in (0.25, 1.0):
        fused = ops.raw_conv_fprop_pooled(x, wq, b, r, 3, False, 128, scale)
        ref = ops.raw_pool(ops.raw_conv_fprop(x, wq, b, r, 3, False, 0, dt, 128, layout).float(), scale)
        assert fused.shape == (2, 128, h // 2, w // 2)
        assert rel_err(fused, ref) < 6e-3          # one bf16 rounding of the pooled value vs four of the full-res ones


def test_res_block_with_fused_downsample_bf16():
    """ResBlock(pool=True) == avg_pool2x2(ResBlock(x)), forward and every gradient (bf16 throughput mode)"""
    g = torch.Generator().manual_seed(77)
    blk = ae.ResBlock(128, 128).to(DEV)
    x = dev(torch.randn(2, 128, 16, 32, generator=g), torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dy = dev(torch.randn(2, 128, 8, 16, generator=g), torch.bfloat16).contiguous(memory_format=torch.channels_last)
    params = list(blk.parameters())
    xa = x.clone().requires_grad_(True)
    ya = blk(xa, pool=True)
    ga = torch.autograd.grad(ya, [xa] + params, dy)
    xb = x.clone().requires_grad_(True)
    yb = ops.avg_pool2x2(blk(xb))
    gb = torch.autograd.grad(yb, [xb] + params, dy)
    assert rel_err(ya, yb) < 6e-3
    for a, b in zip(ga, gb):
        assert rel_err(a, b) < 2e-2


def test_conv_padded_edges_fp32():
    """3-channel image in (padded to 4) and 3-channel reconstruction out (padded to 4) + tanh epilogue."""
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 3, 8, 8, generator=g)
    w1 = torch.randn(32, 3, 3, 3, generator=g) * 0.2
    w2 = torch.randn(3, 32, 3, 3, generator=g) * 0.1
    b2 = torch.randn(3, generator=g) * 0.1
    leaves = [t.clone().requires_grad_(True) for t in (x, w1, w2, b2)]
    yr = torch.tanh(F.conv2d(F.conv2d(leaves[0], leaves[1], None, padding=1), leaves[2], leaves[3], padding=1))
    dy = torch.randn(yr.shape, generator=g)
    gr = torch.autograd.grad(yr, leaves[1:], dy)
    xd = ae._to_internal(dev(x), torch.float32)
    w1d, w2d, b2d = (dev(t).requires_grad_(True) for t in (w1, w2, b2))
    h = ops.conv2d(xd, w1d)
    y = ops.conv2d(h, w2d, b2d, None, False, 1, None)
    assert y.shape == (2, 4, 8, 8)
    close(y[:, :3], yr, rtol=1e-4, atol=1e-5)
    assert float(y.detach()[:, 3].abs().max()) == 0.0
    dyp = F.pad(dev(dy), (0, 0, 0, 0, 0, 1)).contiguous(memory_format=torch.channels_last)
    gd = torch.autograd.grad(y, [w1d, w2d, b2d], dyp)
    for a, r in zip(gd, gr):
        assert rel_err(a, r) < 3e-5


def test_conv_padded_edges_bf16():
    """bf16 throughput mode: 3-channel image in (padded to 8), 3-channel reconstruction out (padded to 8) + tanh,
    on a map wide enough (W % 32 == 0) for the dedicated edge-conv kernels."""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 3, 16, 32, generator=g).bfloat16().float()
    w1 = (torch.randn(128, 3, 3, 3, generator=g) * 0.2).bfloat16().float()
    w2 = (torch.randn(3, 128, 3, 3, generator=g) * 0.05).bfloat16().float()
    b2 = torch.randn(3, generator=g) * 0.1
    leaves = [t.clone().requires_grad_(True) for t in (x, w1, w2, b2)]
    hr = F.conv2d(leaves[0], leaves[1], None, padding=1)
    yr = torch.tanh(F.conv2d(hr, leaves[2], leaves[3], padding=1))
    dy = torch.randn(yr.shape, generator=g).bfloat16().float()
    gr = torch.autograd.grad(yr, leaves[1:], dy)
    xd = ae._to_internal(dev(x), torch.bfloat16)
    w1d, w2d, b2d = (dev(t).requires_grad_(True) for t in (w1, w2, b2))
    h = ops.conv2d(xd, w1d)
    assert rel_err(h, hr) < 6e-3
    y = ops.conv2d(h, w2d, b2d, None, False, 1, None)
    assert y.shape == (2, 8, 16, 32) and y.dtype == torch.bfloat16
    assert rel_err(y[:, :3], yr) < 2e-2
    assert float(y.detach()[:, 3:].abs().max()) == 0.0
    dyp = F.pad(dev(dy), (0, 0, 0, 0, 0, 5)).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    gd = torch.autograd.grad(y, [w1d, w2d, b2d], dyp)
    for a, r in zip(gd, gr):
        assert a.shape == r.shape and rel_err(a, r) < 3e-2


# ------------------------------------------------------------------------------------------ GN / pooling
@pytest.mark.parametrize('tag', ['gn_a', 'gn_b', 'gn_c'])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_group_norm_silu_golden(golden, tag, dtype):
    g = golden('ops')
    x = dev(g[f'{tag}.x'], dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = dev(g[f'{tag}.w']).view(1, -1, 1, 1).requires_grad_(True)
    b = dev(g[f'{tag}.b']).view(1, -1, 1, 1).requires_grad_(True)
    lo = dtype == torch.bfloat16
    close(ops.group_norm_silu(x, w, b, 32, 1e-6, False), g[f'{tag}.gn'], rtol=2e-2 if lo else 1e-4, atol=3e-2 if lo else 2e-5)
    y = ops.group_norm_silu(x, w, b, 32, 1e-6, True)
    close(y, g[f'{tag}.y'], rtol=2e-2 if lo else 1e-4, atol=3e-2 if lo else 2e-5)
    dx, dw, db = torch.autograd.grad(y, [x, w, b], dev(g[f'{tag}.dy'], dtype).contiguous(memory_format=torch.channels_last))
    if lo:
        assert rel_err(dx, T(g[f'{tag}.dx'])) < 2e-2 and rel_err(dw.reshape(-1), T(g[f'{tag}.dw'])) < 2e-2
    else:
        close(dx, g[f'{tag}.dx'], rtol=1e-3, atol=2e-5)
        close(dw.reshape(-1), g[f'{tag}.dw'], rtol=1e-4, atol=2e-5)
        close(db.reshape(-1), g[f'{tag}.db'], rtol=1e-4, atol=2e-5)


def test_group_norm_large_mean_fp32():
    """mean >> std: the E[x^2]-E[x]^2 form must survive (double accumulation)."""
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 64, 32, 32, generator=g) * 0.05 + 7.0
    w, b = torch.ones(64), torch.zeros(64)
    y = ops.group_norm_silu(dev(x).contiguous(memory_format=torch.channels_last), dev(w), dev(b), 32, 1e-6, False)
    close(y, O.group_norm(x.double(), w.double(), b.double()).float(), rtol=2e-3, atol=2e-3)


def test_pool_unpool_golden(golden):
    g = golden('ops')
    x = dev(g['down.x']).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = ops.avg_pool2x2(x)
    close(y, g['down.y'], rtol=1e-6, atol=1e-7)
    close(torch.autograd.grad(y, x, dev(g['down.dy']))[0], g['down.dx'], rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize('tag,cin,cout', [('rb_same', 64, 64), ('rb_proj', 32, 64)])
def test_res_block_golden(golden, tag, cin, cout):
    g = golden('ops')
    m = ae.ResBlock(cin, cout).to(DEV)
    with torch.no_grad():
        for n, p in m.named_parameters():
            p.copy_(dev(g[f'{tag}.p.{n}']))
    x = dev(g[f'{tag}.x']).requires_grad_(True)
    y = m(x)
    close(y, g[f'{tag}.y'], rtol=1e-4, atol=2e-5)
    names = [n for n, _ in m.named_parameters()]
    grads = torch.autograd.grad(y, [x] + [p for _, p in m.named_parameters()], dev(g[f'{tag}.dy']))
    close(grads[0], g[f'{tag}.dx'], rtol=1e-3, atol=3e-5)
    for n, gr in zip(names, grads[1:]):
        assert rel_err(gr, T(g[f'{tag}.g.{n}'])) < 1e-4, n


def test_upsample_golden(golden):
    g = golden('ops')
    m = ae.Upsample(32).to(DEV)
    with torch.no_grad():
        m.conv.weight.copy_(dev(g['up.w']))
        m.conv.bias.copy_(dev(g['up.b']))
    x = dev(g['up.x']).requires_grad_(True)
    y = m(x)
    close(y, g['up.y'], rtol=1e-4, atol=1e-5)
    dx, dw, db = torch.autograd.grad(y, [x, m.conv.weight, m.conv.bias], dev(g['up.dy']))
    close(dx, g['up.dx'], rtol=1e-4, atol=2e-5)
    assert rel_err(dw, T(g['up.dw'])) < 1e-5
    close(db, g['up.db'], rtol=1e-4, atol=1e-5)


# ------------------------------------------------------------------------------------------ StyleGAN2 plugin ops
def test_bias_act_golden(golden):
    g = golden('stylegan_ops')
    gains = {'1.414': float(np.sqrt(2)), '0.707': float(np.sqrt(0.5)), '1.000': 1.0}
    for tag in sorted({k.rsplit('.', 1)[0] for k in g if k.startswith('ba_')}):
        act, gain = tag.split('_')[1], gains[tag.split('_')[2]]
        x = dev(g[f'{tag}.x']).requires_grad_(True)
        b = dev(g[f'{tag}.b']).requires_grad_(True) if f'{tag}.b' in g else None
        y = ops.bias_act(x, b, dim=1, act=act, alpha=0.2 if act == 'lrelu' else None, gain=gain)
        close(y, g[f'{tag}.y'], rtol=1e-6, atol=1e-7)
        gr = torch.autograd.grad(y, [x] + ([b] if b is not None else []), dev(g[f'{tag}.dy']))
        close(gr[0], g[f'{tag}.dx'], rtol=1e-6, atol=1e-7)
        if b is not None:
            close(gr[1], g[f'{tag}.db'], rtol=1e-4, atol=1e-5)


def test_upfirdn2d_golden(golden):
    g = golden('stylegan_ops')
    f = dev(g['uf.f'])
    cases = {'down2_pad1': dict(up=1, down=2, padding=[1, 1, 1, 1], flip_filter=False),
             'filt_pad2': dict(up=1, down=1, padding=[2, 2, 2, 2], flip_filter=False),
             'up2_bwd': dict(up=2, down=1, padding=[2, 1, 2, 1], flip_filter=True),
             'filt_bwd': dict(up=1, down=1, padding=[1, 1, 1, 1], flip_filter=True)}
    for tag, kw in cases.items():
        x = dev(g[f'uf.{tag}.x']).requires_grad_(True)
        y = ops.upfirdn2d(x, f, **kw)
        close(y, g[f'uf.{tag}.y'], rtol=1e-5, atol=1e-6)
        close(torch.autograd.grad(y, x, dev(g[f'uf.{tag}.dy']))[0], g[f'uf.{tag}.dx'], rtol=1e-5, atol=1e-6)


def test_augment_preprocess_vs_oracle_and_identity():
    """fused RandomResizedCrop + HFlip + normalise kernel (base_autoencoder.py:20-22,44-48) for given draws"""
    g = torch.Generator().manual_seed(21)
    images = torch.rand(5, 3, 24, 40, generator=g) * 1.2 - 0.1          # also exercises the clamp
    box = torch.tensor([[0, 0, 40, 24], [3, 2, 20, 20], [10.0, 0, 30, 24], [0, 4, 33, 17], [39, 23, 1, 1]], dtype=torch.float32)
    flip = torch.tensor([0, 1, 0, 1, 1], dtype=torch.int32)
    ref = O.augment_crop_flip(images, box, flip)
    for dt, tol in ((torch.float32, 1e-5), (torch.bfloat16, 8e-3)):
        x, tgt = ops.raw_augment_preprocess(dev(images), dev(box), dev(flip), dt, want_target=True)
        assert x.shape == (5, ops.epc(dt), 24, 40) and float(x[:, 3:].abs().max()) == 0.0
        close(tgt[:, :3], ref, rtol=1e-5, atol=1e-5)
        close(x[:, :3].float(), ref, rtol=tol, atol=tol)
    # the full-image box without flip is exactly the plain preprocess kernel
    plain, _ = ops.raw_preprocess(dev(images[:1]), torch.float32, want_target=False)
    augm, _ = ops.raw_augment_preprocess(dev(images[:1]), dev(box[:1]), dev(flip[:1]), torch.float32, want_target=False)
    assert torch.equal(plain, augm)
    # device-side draws: boxes inside the image, square (ratio 1), area fraction in [0.7, 1]
    b, f = ops.random_crop_params(4096, 256, 256, torch.device(DEV))
    b = b.cpu()
    assert (b[:, 0] >= 0).all() and (b[:, 1] >= 0).all() and (b[:, 0] + b[:, 2] <= 256).all() and (b[:, 1] + b[:, 3] <= 256).all()
    assert (b[:, 2] == b[:, 3]).all()
    frac = (b[:, 2] * b[:, 3]) / 65536.0
    assert 0.69 < float(frac.min()) and float(frac.max()) <= 1.0 and abs(float(frac.mean()) - 0.85) < 0.02
    assert abs(float(f.float().mean()) - 0.5) < 0.05


def test_training_step_with_augmentation_runs():
    torch.manual_seed(0)
    model_mod = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd.model')
    trainer_mod = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd.trainer')
    ae_c = dict(channels=32, num_res_blocks=1, channel_multipliers=(1, 2))
    qc = dict(num_embeddings=64, embedding_dim=16, reinit_every_n_epochs=None, type='standard', params=dict(commitment_cost=0.25))
    tc = dict(lr=1e-4, betas=(0.0, 0.99), eps=1e-8, weight_decay=1e-4, warmup_epochs=None, decay_epochs=None)
    m = model_mod.VQVAE(32, ae_c, qc, None, tc, training_augmentation=True).to(DEV).train()
    tr = trainer_mod.MiniTrainer(num_training_batches=2)
    tr.attach(m)
    x = torch.rand(4, 3, 32, 32, device=DEV)
    l0 = float(tr.train_batch(m, x, 0))
    l1 = float(tr.train_batch(m, x, 1))
    assert np.isfinite(l0) and np.isfinite(l1)
    aug = m.preprocess_batch(x, training=True)
    assert aug.shape == x.shape and float(aug.min()) >= -1.0 and float(aug.max()) <= 1.0


def test_native_library_is_loaded():
    native = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd._native')
    assert native.lib().vqk_arch() == b'gfx950'
    with open('/proc/self/maps') as fh:
        assert any('libvqk.so' in line for line in fh)

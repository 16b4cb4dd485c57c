"""Pins the CPU oracle (oracle/) against the golden vectors captured from the reference."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import vqvae_oracle as O
from oracle import vq_c

T = torch.from_numpy


def close(a, b, rtol=1e-5, atol=1e-6):
    np.testing.assert_allclose(a.detach().numpy() if isinstance(a, torch.Tensor) else a, b, rtol=rtol, atol=atol)


@pytest.mark.parametrize('tag', ['gn_a', 'gn_b', 'gn_c'])
def test_group_norm_silu(golden, tag):
    g = golden('ops')
    x = T(g[f'{tag}.x']).requires_grad_(True)
    w = T(g[f'{tag}.w']).requires_grad_(True)
    b = T(g[f'{tag}.b']).requires_grad_(True)
    gn = O.group_norm(x, w, b)
    close(gn, g[f'{tag}.gn'])
    y = F.silu(gn)
    close(y, g[f'{tag}.y'])
    dx, dw, db = torch.autograd.grad(y, [x, w, b], T(g[f'{tag}.dy']))
    close(dx, g[f'{tag}.dx'], atol=1e-5)
    close(dw, g[f'{tag}.dw'], rtol=1e-4, atol=1e-5)
    close(db, g[f'{tag}.db'], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize('tag', ['rb_same', 'rb_proj'])
def test_res_block(golden, tag):
    g = golden('ops')
    p = {k[len(tag) + 3:]: T(v).requires_grad_(True) for k, v in g.items() if k.startswith(tag + '.p.')}
    x = T(g[f'{tag}.x']).requires_grad_(True)
    y = O.res_block(x, p, '')
    close(y, g[f'{tag}.y'], atol=1e-5)
    names = sorted(p)
    grads = torch.autograd.grad(y, [x] + [p[n] for n in names], T(g[f'{tag}.dy']))
    close(grads[0], g[f'{tag}.dx'], rtol=1e-4, atol=1e-5)
    for n, gr in zip(names, grads[1:]):
        close(gr, g[f'{tag}.g.{n}'], rtol=1e-4, atol=2e-5)


def test_down_up(golden):
    g = golden('ops')
    x = T(g['down.x']).requires_grad_(True)
    y = O.downsample(x)
    close(y, g['down.y'])
    close(torch.autograd.grad(y, x, T(g['down.dy']))[0], g['down.dx'])
    x = T(g['up.x']).requires_grad_(True)
    w = T(g['up.w']).requires_grad_(True)
    b = T(g['up.b']).requires_grad_(True)
    y = O.upsample(x, w, b)
    close(y, g['up.y'], atol=1e-5)
    dx, dw, db = torch.autograd.grad(y, [x, w, b], T(g['up.dy']))
    close(dx, g['up.dx'], rtol=1e-4, atol=1e-5)
    close(dw, g['up.dw'], rtol=1e-4, atol=1e-5)
    close(db, g['up.db'], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize('tag', ['std_a', 'std_b', 'std_c'])
def test_vq_standard(golden, tag):
    g = golden('vq')
    z = T(g[f'{tag}.z']).requires_grad_(True)
    e = T(g[f'{tag}.e']).requires_grad_(True)
    q, idx, loss = O.vq_standard(z, e, 0.25)
    assert np.array_equal(idx.numpy(), g[f'{tag}.idx'])
    assert np.array_equal(idx.numpy(), g[f'{tag}.codes'])
    close(q, g[f'{tag}.q'])
    close(loss, g[f'{tag}.loss'])
    dz, de = torch.autograd.grad([q, loss], [z, e], [T(g[f'{tag}.dq']), torch.tensor(1.0)])
    close(dz, g[f'{tag}.dz'])
    close(de, g[f'{tag}.de'])
    # canonical-order C oracle agrees with the reference on every index (ties included)
    fz = g[f'{tag}.z'].transpose(0, 2, 3, 1).reshape(-1, g[f'{tag}.z'].shape[1])
    cidx, _, _, _ = vq_c.assign(fz, g[f'{tag}.e'], assoc=0)
    assert np.array_equal(cidx, g[f'{tag}.idx'].reshape(-1))


def test_vq_ties_pick_lower_index(golden):
    g = golden('vq')
    idx = g['std_a.idx'].reshape(-1)
    k = g['std_a.e'].shape[0]
    assert not np.isin(idx, [k // 2, k - 1]).any()      # duplicates of row 3 must never win


def test_vq_ema_trajectory(golden):
    g = golden('vq')
    cb, cnt, w = T(g['ema.e0']), T(g['ema.c0']), T(g['ema.w0'])
    for s in range(3):
        z = T(g[f'ema.z{s}']).requires_grad_(True)
        q, idx, loss, cnt, w, cb = O.vq_ema(z, cb, cnt, w, 0.25, 0.95, 1e-5)
        assert np.array_equal(idx.numpy(), g[f'ema.idx{s}'])
        close(q, g[f'ema.q{s}'])
        close(loss, g[f'ema.loss{s}'])
        close(cnt, g[f'ema.count{s}'])
        close(w, g[f'ema.weight{s}'])
        close(cb, g[f'ema.cb{s}'], rtol=1e-5)
        dz, = torch.autograd.grad([q, loss], [z], [T(g[f'ema.dq{s}']), torch.tensor(1.0)])
        close(dz, g[f'ema.dz{s}'])


def test_vq_entropy(golden):
    g = golden('vq')
    z = T(g['ent.z']).requires_grad_(True)
    e = T(g['ent.e']).requires_grad_(True)
    q, idx, loss = O.vq_entropy(z, e, 0.25, 0.1, 0.01)
    assert np.array_equal(idx.numpy(), g['ent.idx'])
    close(q, g['ent.q'])
    close(loss, g['ent.loss'], rtol=1e-5)
    dz, de = torch.autograd.grad([q, loss], [z, e], [T(g['ent.dq']), torch.tensor(1.0)])
    close(dz, g['ent.dz'], rtol=1e-4, atol=1e-6)
    close(de, g['ent.de'], rtol=1e-4, atol=1e-6)
    fz = g['ent.z'].transpose(0, 2, 3, 1).reshape(-1, g['ent.z'].shape[1])
    cidx, _, _, _ = vq_c.assign(fz, g['ent.e'], assoc=1)
    assert np.array_equal(cidx, g['ent.idx'].reshape(-1))


def test_vq_gumbel(golden):
    g = golden('vq')
    x = T(g['gum.x']).requires_grad_(True)
    e, w, b = (T(g[k]).requires_grad_(True) for k in ('gum.e', 'gum.w', 'gum.b'))
    q, idx, loss = O.vq_gumbel(x, e, w, b, 0.7, 5e-4, T(g['gum.noise']))
    assert np.array_equal(idx.numpy(), g['gum.idx'])
    close(q, g['gum.q'])
    close(loss, g['gum.loss'])
    gr = torch.autograd.grad([q, loss], [x, e, w, b], [T(g['gum.dq']), torch.tensor(1.0)])
    for a, k in zip(gr, ('gum.dx', 'gum.de', 'gum.dw', 'gum.db')):
        close(a, g[k], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize('tag', ['a', 'b'])
def test_vq_entropy_argmax(golden, tag):
    """ent_loss_type='argmax' (vector_quantizers.py:311-315), vectors from the reference's own module"""
    g = golden('vq_entropy_argmax')
    z = T(g[f'{tag}.z']).requires_grad_(True)
    e = T(g[f'{tag}.e']).requires_grad_(True)
    q, idx, loss = O.vq_entropy(z, e, 0.25, 0.1, float(g[f'{tag}.temp']), 'argmax')
    assert np.array_equal(idx.numpy(), g[f'{tag}.idx'])
    close(loss, g[f'{tag}.loss'], rtol=1e-5)
    dz, de = torch.autograd.grad([q, loss], [z, e], [T(g[f'{tag}.dq']), torch.tensor(1.0)])
    close(dz, g[f'{tag}.dz'], rtol=1e-4, atol=1e-6)
    close(de, g[f'{tag}.de'], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize('tag,assoc', [('n1', 0), ('n036', 0), ('ent', 1)])
def test_vq_large_indices_canonical_order(golden, tag, assoc):
    """BASELINE shape: the canonical-order C oracle reproduces every reference index."""
    g = golden('vq_large')
    gen = torch.Generator().manual_seed(int(g[f'{tag}.seed']))
    z = torch.randn(32, 256, 16, 16, generator=gen) * float(g[f'{tag}.scale'])
    e = (torch.rand(1024, 256, generator=gen) * 2 - 1) / 1024
    fz = z.permute(0, 2, 3, 1).reshape(-1, 256)
    if tag != 'n1':
        e = fz[torch.randperm(8192, generator=gen)[:1024]] + 0.01 * torch.randn(1024, 256, generator=gen)
    ref = g[f'{tag}.idx'].astype(np.int64).reshape(-1)
    n = 2048                                # bounded: the scalar C loop is ~1 s per 2048 rows
    cidx, _, _, _ = vq_c.assign(fz[:n].numpy(), e.numpy(), assoc=assoc)
    assert np.array_equal(cidx, ref[:n])
    tidx = torch.argmin(O.distances_std(fz, e) if assoc == 0 else O.distances_entropy(fz, e), dim=1)
    assert np.array_equal(tidx.numpy(), ref)


def _load_step(golden, qtype):
    base = golden('train_step_standard')
    g = dict(golden(f'train_step_{qtype}'))
    p = {k: T(v) for k, v in base.items() if k.startswith(('encoder.', 'decoder.'))}
    src = base if qtype == 'standard' else g
    p.update({k: T(v) for k, v in src.items() if k.startswith('quantizer.')})
    return base, g, p


@pytest.mark.parametrize('qtype', ['standard', 'ema', 'entropy'])
def test_train_step(golden, qtype):
    base, g, p = _load_step(golden, qtype)
    qparams = {'standard': dict(commitment_cost=0.25),
               'ema': dict(commitment_cost=0.25, decay=0.95, epsilon=1e-5),
               'entropy': dict(commitment_cost=0.25, ent_loss_ratio=0.1, ent_temperature=0.01)}[qtype]
    buffers = None
    if qtype == 'ema':
        buffers = dict(ema_count=p.pop('quantizer.ema_count'), ema_weight=p.pop('quantizer.ema_weight'))
    r = O.train_step_mse(T(base['images']), p, 1, 2, qtype, qparams, buffers)
    assert np.array_equal(r['idx'].numpy(), g['out.idx'])
    close(r['z'], g['out.z'], rtol=1e-4, atol=1e-5)
    close(r['recon'], g['out.recon'], rtol=1e-4, atol=1e-5)
    close(r['q_loss'], g['out.q_loss'], rtol=1e-5)
    close(r['l2'], g['out.l2'], rtol=1e-5)
    n_checked = 0
    for k, v in g.items():
        if k.startswith('grad.'):
            close(r['grads'][k[5:]], v, rtol=2e-3, atol=2e-6)
            n_checked += 1
    assert n_checked >= 10
    if qtype == 'ema':
        close(r['extra']['ema_count'], g['after.ema_count'])
        close(r['extra']['ema_weight'], g['after.ema_weight'])
        close(r['extra']['codebook'], g['after.codebook.weight'], rtol=1e-5)


def test_adamw_groups_and_step(golden):
    base, g, p = _load_step(golden, 'standard')
    names = [k[5:] for k in g if k.startswith('grad.')]
    decay, no_decay = O.decay_split(names)
    assert sorted(decay) == sorted(g['decay_names'].tolist())
    assert len(decay) + len(no_decay) == len(names)
    for n in names:
        wd = 1e-4 if n in decay else 0.0
        newp, _, _ = O.adamw_step(p[n], T(g['grad.' + n]), torch.zeros_like(p[n]), 1, 1e-4, 0.0, 0.99, 1e-8, wd)
        close(newp, g['stepped.' + n], rtol=1e-6, atol=1e-7)


def test_bias_act_and_upfirdn2d(golden):
    g = golden('stylegan_ops')
    for tag in sorted({k.rsplit('.', 1)[0] for k in g if k.startswith('ba_')}):
        act = tag.split('_')[1]
        gain = {'1.414': float(np.sqrt(2)), '0.707': float(np.sqrt(0.5)), '1.000': 1.0}[tag.split('_')[2]]
        x = T(g[f'{tag}.x']).requires_grad_(True)
        b = T(g[f'{tag}.b']).requires_grad_(True) if f'{tag}.b' in g else None
        y = O.bias_act(x, b, 1, act, 0.2, gain)
        close(y, g[f'{tag}.y'], rtol=1e-5)
        gr = torch.autograd.grad(y, [x] + ([b] if b is not None else []), T(g[f'{tag}.dy']))
        close(gr[0], g[f'{tag}.dx'], rtol=1e-5)
        if b is not None:
            close(gr[1], g[f'{tag}.db'], rtol=1e-4, atol=1e-5)
    f = T(g['uf.f'])
    cases = {'down2_pad1': dict(up=(1, 1), down=(2, 2), pad=(1, 1, 1, 1), flip_filter=False),
             'filt_pad2': dict(up=(1, 1), down=(1, 1), pad=(2, 2, 2, 2), flip_filter=False),
             'up2_bwd': dict(up=(2, 2), down=(1, 1), pad=(2, 1, 2, 1), flip_filter=True),
             'filt_bwd': dict(up=(1, 1), down=(1, 1), pad=(1, 1, 1, 1), flip_filter=True)}
    for tag, kw in cases.items():
        x = T(g[f'uf.{tag}.x']).requires_grad_(True)
        y = O.upfirdn2d(x, f, **kw)
        close(y, g[f'uf.{tag}.y'], rtol=1e-5, atol=1e-6)
        close(torch.autograd.grad(y, x, T(g[f'uf.{tag}.dy']))[0], g[f'uf.{tag}.dx'], rtol=1e-5, atol=1e-6)


def test_augment_restatement_identity_and_flip():
    """oracle.augment_crop_flip: the full-image box is the plain normalisation; a flipped full box mirrors it"""
    g = torch.Generator().manual_seed(3)
    x = torch.rand(2, 3, 9, 13, generator=g)
    box = torch.tensor([[0, 0, 13, 9], [0, 0, 13, 9]], dtype=torch.float32)
    out = O.augment_crop_flip(x, box, torch.tensor([0, 1]))
    close(out[0], (x[0] - 0.5) / 0.5, rtol=1e-6, atol=1e-6)
    close(out[1], ((x[1] - 0.5) / 0.5).flip(-1), rtol=1e-6, atol=1e-6)

"""Generate the golden vectors under tests/golden/ by running the REFERENCE's own modules on CPU.

Run in the build container only (the reference checkout is not available on the GPU box):

    PYTHONPATH=/root/reference python tests/golden/make_golden.py

Only inputs / expected outputs are written (``.npz``); no reference source is stored.
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

REF = os.environ.get('VQK_REFERENCE', '/root/reference')
sys.path.insert(0, REF)
OUT = os.path.dirname(os.path.abspath(__file__))

from vqvae.modules.autoencoder import GroupNorm, ResBlock, Downsample, Upsample, Encoder, Decoder  # noqa: E402
from vqvae.modules.vector_quantizers import (VectorQuantizer, EMAVectorQuantizer, EntropyVectorQuantizer,  # noqa: E402
                                             GumbelVectorQuantizer)
from vqvae.modules.loss.stylegan2_discriminator.utils.ops import bias_act as ref_bias_act  # noqa: E402
from vqvae.modules.loss.stylegan2_discriminator.utils.ops import upfirdn2d as ref_upfirdn2d  # noqa: E402

torch.set_num_threads(4)


def npy(t):
    return t.detach().cpu().numpy().copy()


def sd(mod, pre=''):
    return {pre + k: npy(v) for k, v in mod.state_dict().items()}


def save(name, **arrs):
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **arrs)
    print(f'{name}: {os.path.getsize(path) / 1024:.1f} KiB, {len(arrs)} arrays')


# ---------------------------------------------------------------- per-op vectors
def gen_ops():
    g = torch.Generator().manual_seed(1234)
    out = {}
    for tag, (b, c, h, w) in {'gn_a': (2, 32, 5, 7), 'gn_b': (2, 128, 4, 4), 'gn_c': (1, 64, 8, 8)}.items():
        m = GroupNorm(32, c)
        with torch.no_grad():
            m.weight.copy_(torch.randn(m.weight.shape, generator=g) * 0.3 + 1)
            m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.2)
        x = (torch.randn(b, c, h, w, generator=g) * 1.7 + 0.3).requires_grad_(True)
        y = F.silu(m(x))
        dy = torch.randn(y.shape, generator=g)
        dx, dw, db = torch.autograd.grad(y, [x, m.weight, m.bias], dy)
        out.update({f'{tag}.x': npy(x), f'{tag}.w': npy(m.weight).reshape(-1), f'{tag}.b': npy(m.bias).reshape(-1),
                    f'{tag}.gn': npy(m(x)), f'{tag}.y': npy(y), f'{tag}.dy': npy(dy), f'{tag}.dx': npy(dx),
                    f'{tag}.dw': npy(dw).reshape(-1), f'{tag}.db': npy(db).reshape(-1)})
    for tag, (cin, cout, hw) in {'rb_same': (64, 64, 8), 'rb_proj': (32, 64, 6)}.items():
        torch.manual_seed(7)
        m = ResBlock(cin, cout)
        with torch.no_grad():
            for p in m.parameters():
                if p.ndim == 4 and p.shape[0] == 1:
                    p.add_(torch.randn(p.shape, generator=g) * 0.1)
        x = torch.randn(2, cin, hw, hw, generator=g).requires_grad_(True)
        y = m(x)
        dy = torch.randn(y.shape, generator=g)
        params = list(m.named_parameters())
        grads = torch.autograd.grad(y, [x] + [p for _, p in params], dy)
        out.update({f'{tag}.x': npy(x), f'{tag}.y': npy(y), f'{tag}.dy': npy(dy), f'{tag}.dx': npy(grads[0])})
        out.update({f'{tag}.p.{n}': npy(p) for n, p in params})
        out.update({f'{tag}.g.{n}': npy(gr) for (n, _), gr in zip(params, grads[1:])})
    x = torch.randn(2, 32, 6, 10, generator=g).requires_grad_(True)
    y = Downsample()(x)
    dy = torch.randn(y.shape, generator=g)
    out.update({'down.x': npy(x), 'down.y': npy(y), 'down.dy': npy(dy), 'down.dx': npy(torch.autograd.grad(y, x, dy)[0])})
    torch.manual_seed(11)
    up = Upsample(32)
    x = torch.randn(2, 32, 5, 3, generator=g).requires_grad_(True)
    y = up(x)
    dy = torch.randn(y.shape, generator=g)
    gx, gw, gb = torch.autograd.grad(y, [x, up.conv.weight, up.conv.bias], dy)
    out.update({'up.x': npy(x), 'up.w': npy(up.conv.weight), 'up.b': npy(up.conv.bias), 'up.y': npy(y), 'up.dy': npy(dy),
                'up.dx': npy(gx), 'up.dw': npy(gw), 'up.db': npy(gb)})
    save('ops', **out)


# ---------------------------------------------------------------- quantizers
def _vq_inputs(g, n_img, d, hw, k, scale, dup=True):
    z = torch.randn(n_img, d, hw, hw, generator=g) * scale
    e = (torch.rand(k, d, generator=g) * 2 - 1) / k
    if dup:                               # duplicate code rows -> exact ties, lower index must win
        e[k // 2] = e[3]
        e[k - 1] = e[3]
    return z, e


def gen_vq():
    g = torch.Generator().manual_seed(4321)
    out = {}
    # standard: small + trained-like codebook + scaled inputs
    for tag, scale, trained in (('std_a', 1.0, False), ('std_b', 0.36, True), ('std_c', 0.01, False)):
        z, e = _vq_inputs(g, 2, 16, 8, 64, scale)
        if trained:
            fz = z.permute(0, 2, 3, 1).reshape(-1, 16)
            e = fz[torch.randperm(fz.shape[0], generator=g)[:64]] + 0.01 * torch.randn(64, 16, generator=g)
            e[40] = e[5]
        q = VectorQuantizer(64, 16, 0.25)
        with torch.no_grad():
            q.codebook.weight.copy_(e)
        zz = z.clone().requires_grad_(True)
        qz, idx, loss = q(zz)
        dq = torch.randn(qz.shape, generator=g)
        dz, de = torch.autograd.grad([qz, loss], [zz, q.codebook.weight], [dq, torch.tensor(1.0)])
        out.update({f'{tag}.z': npy(z), f'{tag}.e': npy(e), f'{tag}.q': npy(qz), f'{tag}.idx': npy(idx),
                    f'{tag}.loss': npy(loss), f'{tag}.dq': npy(dq), f'{tag}.dz': npy(dz), f'{tag}.de': npy(de),
                    f'{tag}.codes': npy(q.vec_to_codes(z))})
    # EMA: three training steps, state after each
    z0, e = _vq_inputs(g, 4, 16, 4, 32, 1.0, dup=False)
    q = EMAVectorQuantizer(32, 16, 0.25, 0.95, 1e-5)
    with torch.no_grad():
        q.codebook.weight.copy_(e)
    out.update({'ema.e0': npy(e), 'ema.w0': npy(q.ema_weight), 'ema.c0': npy(q.ema_count)})
    q.train()
    for s in range(3):
        z = (torch.randn(4, 16, 4, 4, generator=g) * 0.5).requires_grad_(True)
        qz, idx, loss = q(z)
        dq = torch.randn(qz.shape, generator=g)
        dz, = torch.autograd.grad([qz, loss], [z], [dq, torch.tensor(1.0)])
        out.update({f'ema.z{s}': npy(z), f'ema.q{s}': npy(qz), f'ema.idx{s}': npy(idx), f'ema.loss{s}': npy(loss),
                    f'ema.dq{s}': npy(dq), f'ema.dz{s}': npy(dz), f'ema.count{s}': npy(q.ema_count),
                    f'ema.weight{s}': npy(q.ema_weight), f'ema.cb{s}': npy(q.codebook.weight)})
    # entropy (softmax) fwd/bwd
    z, e = _vq_inputs(g, 2, 16, 8, 64, 0.3)
    q = EntropyVectorQuantizer(64, 16, 0.1, 0.01, 'softmax', 0.25)
    with torch.no_grad():
        q.codebook.weight.copy_(e)
    zz = z.clone().requires_grad_(True)
    qz, idx, loss = q(zz)
    dq = torch.randn(qz.shape, generator=g)
    dz, de = torch.autograd.grad([qz, loss], [zz, q.codebook.weight], [dq, torch.tensor(1.0)])
    out.update({'ent.z': npy(z), 'ent.e': npy(e), 'ent.q': npy(qz), 'ent.idx': npy(idx), 'ent.loss': npy(loss),
                'ent.dq': npy(dq), 'ent.dz': npy(dz), 'ent.de': npy(de)})
    # gumbel with injected noise: reseed right before forward, the first RNG draw is exponential_()
    k, d = 32, 8
    q = GumbelVectorQuantizer(k, d, False, 0.7, 5e-4)
    q.init_codebook()
    q.train()
    x = torch.randn(2, k, 4, 4, generator=g).requires_grad_(True)
    torch.manual_seed(99)
    noise = torch.empty(2, k, 4, 4).exponential_()
    torch.manual_seed(99)
    qz, idx, loss = q(x)
    dq = torch.randn(qz.shape, generator=g)
    gr = torch.autograd.grad([qz, loss], [x, q.codebook.weight, q.x_to_logits.weight, q.x_to_logits.bias],
                             [dq, torch.tensor(1.0)])
    out.update({'gum.x': npy(x), 'gum.e': npy(q.codebook.weight), 'gum.w': npy(q.x_to_logits.weight),
                'gum.b': npy(q.x_to_logits.bias), 'gum.noise': npy(noise), 'gum.q': npy(qz), 'gum.idx': npy(idx),
                'gum.loss': npy(loss), 'gum.dq': npy(dq), 'gum.dx': npy(gr[0]), 'gum.de': npy(gr[1]),
                'gum.dw': npy(gr[2]), 'gum.db': npy(gr[3])})
    save('vq', **out)


def gen_vq_large():
    """BASELINE shape (N=8192, K=1024, D=256): inputs are regenerated from seeds, outputs are the
    reference indices (int16), a usage histogram and the loss."""
    out = {}
    for tag, seed, scale, assoc in (('n1', 1234, 1.0, 'std'), ('n036', 1235, 0.36, 'std'), ('ent', 1236, 1.0, 'ent')):
        g = torch.Generator().manual_seed(seed)
        z = torch.randn(32, 256, 16, 16, generator=g) * scale
        e = (torch.rand(1024, 256, generator=g) * 2 - 1) / 1024
        if tag != 'n1':                   # trained-like codebook: rows sampled from z (+ noise)
            fz = z.permute(0, 2, 3, 1).reshape(-1, 256)
            e = fz[torch.randperm(8192, generator=g)[:1024]] + 0.01 * torch.randn(1024, 256, generator=g)
        q = VectorQuantizer(1024, 256, 0.25) if assoc == 'std' else EntropyVectorQuantizer(1024, 256)
        with torch.no_grad():
            q.codebook.weight.copy_(e)
            idx = q.vec_to_codes(z)
        out.update({f'{tag}.seed': np.int64(seed), f'{tag}.scale': np.float32(scale),
                    f'{tag}.idx': npy(idx).astype(np.int16)})
    save('vq_large', **out)


# ---------------------------------------------------------------- tiny autoencoder + train step
TINY = dict(channels=32, num_res_blocks=1, channel_multipliers=(1, 2), embedding_dim=16, num_embeddings=64)


def gen_train_step():
    torch.manual_seed(1234)
    enc = Encoder(TINY['channels'], TINY['num_res_blocks'], TINY['channel_multipliers'], TINY['embedding_dim'])
    dec = Decoder(TINY['channels'], TINY['num_res_blocks'], TINY['channel_multipliers'], TINY['embedding_dim'])
    g = torch.Generator().manual_seed(1234)
    with torch.no_grad():                 # make GroupNorm affine non-trivial
        for m in list(enc.modules()) + list(dec.modules()):
            if isinstance(m, GroupNorm):
                m.weight.add_(torch.randn(m.weight.shape, generator=g) * 0.1)
                m.bias.add_(torch.randn(m.bias.shape, generator=g) * 0.1)
    images = torch.rand(4, 3, 32, 32, generator=g)
    x = (images.clamp(0, 1) - 0.5) / 0.5
    for qtype in ('standard', 'ema', 'entropy'):
        torch.manual_seed(77)
        if qtype == 'standard':
            q = VectorQuantizer(TINY['num_embeddings'], TINY['embedding_dim'], 0.25)
        elif qtype == 'ema':
            q = EMAVectorQuantizer(TINY['num_embeddings'], TINY['embedding_dim'], 0.25, 0.95, 1e-5)
        else:
            q = EntropyVectorQuantizer(TINY['num_embeddings'], TINY['embedding_dim'], 0.1, 0.01, 'softmax', 0.25)
        q.init_codebook()
        with torch.no_grad():             # spread the codebook so several codes are used
            q.codebook.weight.mul_(TINY['num_embeddings'] * 0.5)
        q.train()
        full = qtype == 'standard'        # ema/entropy reuse train_step_standard's images + enc/dec weights
        out = {}
        if full:
            out['images'] = npy(images)
            out.update(sd(enc, 'encoder.'))
            out.update(sd(dec, 'decoder.'))
        out.update({'quantizer.' + k: npy(v) for k, v in q.state_dict().items()})
        z = enc(x)
        qz, idx, ql = q(z)
        recon = dec(qz)
        l2 = F.mse_loss(recon, x)
        loss = ql + l2
        named = ([('encoder.' + n, p) for n, p in enc.named_parameters()]
                 + [('decoder.' + n, p) for n, p in dec.named_parameters()]
                 + [('quantizer.' + n, p) for n, p in q.named_parameters() if p.requires_grad])
        grads = torch.autograd.grad(loss, [p for _, p in named])
        out.update({'out.z': npy(z), 'out.recon': npy(recon), 'out.idx': npy(idx), 'out.q_loss': npy(ql),
                    'out.l2': npy(l2), 'out.loss': npy(loss)})
        keep = ('quantizer.', 'encoder.conv_in', 'encoder.conv_out', 'encoder.norm', 'encoder.blocks.0.',
                'decoder.conv_in', 'decoder.conv_out', 'decoder.norm', 'decoder.blocks.3.')
        out.update({'grad.' + n: npy(gr) for (n, _), gr in zip(named, grads) if full or n.startswith(keep)})
        if qtype == 'ema':
            out.update({'after.' + k: npy(v) for k, v in q.state_dict().items()})
        # one AdamW step with the reference's two groups (model.py:419-428) and yaml hyper-parameters
        decay = [p for n, p in named if p.ndim == 4 and p.shape[0] != 1]
        no_decay = [p for n, p in named if not (p.ndim == 4 and p.shape[0] != 1)]
        opt = torch.optim.AdamW([{'params': decay, 'weight_decay': 1e-4}, {'params': no_decay, 'weight_decay': 0.0}],
                                lr=1e-4, betas=(0.0, 0.99), eps=1e-8, weight_decay=1e-4)
        for (n, p), gr in zip(named, grads):
            p.grad = gr
        opt.step()
        if full:
            out.update({'stepped.' + n: npy(p) for n, p in named})
        out['decay_names'] = np.array([n for n, p in named if p.ndim == 4 and p.shape[0] != 1])
        save(f'train_step_{qtype}', **out)
        if full:
            saved = {n: out[n] for n, _ in named if not n.startswith('quantizer.')}
        with torch.no_grad():             # restore for the next quantizer type
            for (n, p) in named:
                if not n.startswith('quantizer.'):
                    p.copy_(torch.from_numpy(saved[n]))


# ---------------------------------------------------------------- StyleGAN2 custom ops (ref twins)
def gen_stylegan_ops():
    g = torch.Generator().manual_seed(5)
    out = {}
    for act, gain in (('lrelu', float(np.sqrt(2))), ('lrelu', 1.0), ('linear', float(np.sqrt(0.5)))):
        tag = f'ba_{act}_{gain:.3f}'
        x = torch.randn(2, 6, 5, 7, generator=g).requires_grad_(True)
        b = torch.randn(6, generator=g).requires_grad_(True) if act == 'lrelu' else None
        y = ref_bias_act._bias_act_ref(x, b, dim=1, act=act, alpha=0.2 if act == 'lrelu' else None, gain=gain)
        dy = torch.randn(y.shape, generator=g)
        gr = torch.autograd.grad(y, [x] + ([b] if b is not None else []), dy)
        out.update({f'{tag}.x': npy(x), f'{tag}.y': npy(y), f'{tag}.dy': npy(dy), f'{tag}.dx': npy(gr[0])})
        if b is not None:
            out.update({f'{tag}.b': npy(b), f'{tag}.db': npy(gr[1])})
    f = ref_upfirdn2d.setup_filter([1, 3, 3, 1])
    out['uf.f'] = npy(f)
    cases = {'down2_pad1': dict(up=1, down=2, padding=[1, 1, 1, 1], flip_filter=False),
             'filt_pad2': dict(up=1, down=1, padding=[2, 2, 2, 2], flip_filter=False),
             'up2_bwd': dict(up=2, down=1, padding=[2, 1, 2, 1], flip_filter=True),
             'filt_bwd': dict(up=1, down=1, padding=[1, 1, 1, 1], flip_filter=True)}
    for tag, kw in cases.items():
        x = torch.randn(2, 5, 9, 11, generator=g).requires_grad_(True)
        y = ref_upfirdn2d._upfirdn2d_ref(x, f, gain=1, **kw)
        dy = torch.randn(y.shape, generator=g)
        dx, = torch.autograd.grad(y, x, dy)
        out.update({f'uf.{tag}.x': npy(x), f'uf.{tag}.y': npy(y), f'uf.{tag}.dy': npy(dy), f'uf.{tag}.dx': npy(dx)})
    save('stylegan_ops', **out)


def gen_vq_entropy_argmax():
    """Entropy quantizer with ent_loss_type='argmax' (straight-through one-hot targets, vector_quantizers.py:311-315)."""
    g = torch.Generator().manual_seed(4321)
    out = {}
    for tag, (n_img, d, hw, k, scale, temp) in {'a': (2, 16, 8, 64, 0.3, 0.01), 'b': (3, 32, 4, 128, 1.0, 0.5)}.items():
        z, e = _vq_inputs(g, n_img, d, hw, k, scale)
        q = EntropyVectorQuantizer(k, d, 0.1, temp, 'argmax', 0.25)
        with torch.no_grad():
            q.codebook.weight.copy_(e)
        zz = z.clone().requires_grad_(True)
        qz, idx, loss = q(zz)
        dq = torch.randn(qz.shape, generator=g)
        dz, de = torch.autograd.grad([qz, loss], [zz, q.codebook.weight], [dq, torch.tensor(1.0)])
        out.update({f'{tag}.z': npy(z), f'{tag}.e': npy(e), f'{tag}.q': npy(qz), f'{tag}.idx': npy(idx),
                    f'{tag}.loss': npy(loss), f'{tag}.dq': npy(dq), f'{tag}.dz': npy(dz), f'{tag}.de': npy(de),
                    f'{tag}.temp': np.float32(temp)})
    save('vq_entropy_argmax', **out)


if __name__ == '__main__' and 'entropy_argmax' in sys.argv[1:]:
    gen_vq_entropy_argmax()
    sys.exit(0)

if __name__ == '__main__' and 'gan' not in sys.argv[1:]:
    gen_ops()
    gen_vq()
    gen_vq_large()
    gen_train_step()
    gen_stylegan_ops()


# ---------------------------------------------------------------- VQ-GAN loss path
def _stub_torchvision():
    """torchvision is not installed: provide just enough for `vqvae.modules.loss` to import.  vgg16().features is
    a hand-assembled cfg-"D" Sequential with PyTorch-default (seeded) weights -- pretrained values stay unpinned."""
    import types
    tv = types.ModuleType('torchvision')
    models = types.ModuleType('torchvision.models')

    def vgg16(weights=None):
        cfg = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512, 'M']
        layers, cin = [], 3
        for v in cfg:
            if v == 'M':
                layers.append(torch.nn.MaxPool2d(2, 2))
            else:
                layers += [torch.nn.Conv2d(cin, v, 3, padding=1), torch.nn.ReLU(inplace=True)]
                cin = v
        m = types.SimpleNamespace()
        m.features = torch.nn.Sequential(*layers)
        return m
    models.vgg16 = vgg16
    models.VGG16_Weights = types.SimpleNamespace(DEFAULT=None)
    tv.models = models
    sys.modules['torchvision'] = tv
    sys.modules['torchvision.models'] = models


def gen_gan():
    _stub_torchvision()
    from vqvae.modules.loss.stylegan2_discriminator.discriminator import Discriminator
    from vqvae.modules.loss import loss as ref_loss
    from vqvae.modules.loss.lpips_pytorch.modules import lpips as ref_lpips
    out = {}
    g = torch.Generator().manual_seed(21)
    # discriminator: 32x32, 289k parameters
    torch.manual_seed(5)
    d = Discriminator(32, channel_base=1024, channel_max=64)
    with torch.no_grad():
        for n_, p in d.named_parameters():
            if n_.endswith('bias'):
                p.add_(torch.randn(p.shape, generator=g) * 0.1)
    x = (torch.randn(4, 3, 32, 32, generator=g) * 0.5).requires_grad_(True)
    logits = d(x)
    r = torch.randn(4, 1, generator=g)
    named = list(d.named_parameters())
    grads = torch.autograd.grad((logits * r).sum(), [x] + [p for _, p in named])
    out.update({'d.' + k: npy(v) for k, v in d.state_dict().items()})
    out.update({'d_in.x': npy(x), 'd_in.r': npy(r), 'd_out.logits': npy(logits), 'd_out.dx': npy(grads[0])})
    out.update({'d_grad.' + n_: npy(gr) for (n_, _), gr in zip(named, grads[1:])})
    # R1 regularisation (loss.py:98-112): second-order through the whole discriminator
    xr = x.detach().clone().requires_grad_(True)
    lg = d(xr)
    gimg, = torch.autograd.grad(lg.sum(), xr, create_graph=True)
    r1 = 10.0 * gimg.pow(2).view(gimg.shape[0], -1).sum(1).mean()
    r1_grads = torch.autograd.grad(r1, [p for _, p in named], allow_unused=True)
    out.update({'r1.value': npy(r1), 'r1.gimg': npy(gimg)})
    out.update({'r1_grad.' + n_: npy(gr) for (n_, _), gr in zip(named, r1_grads) if gr is not None})
    # GAN losses
    lr_, lf_ = torch.randn(8, 1, generator=g) * 2, torch.randn(8, 1, generator=g) * 2
    for lt in ('hinge', 'non-saturating'):
        a, b = lr_.clone().requires_grad_(True), lf_.clone().requires_grad_(True)
        gl = ref_loss.generator_loss(b, lt)
        out[f'gan.{lt}.g'] = npy(gl)
        out[f'gan.{lt}.g_dfake'] = npy(torch.autograd.grad(gl, b)[0])
        dl = ref_loss.discriminator_loss(a, b, lt)
        da, db = torch.autograd.grad(dl, [a, b])
        out.update({f'gan.{lt}.d': npy(dl), f'gan.{lt}.d_dreal': npy(da), f'gan.{lt}.d_dfake': npy(db)})
    out.update({'gan.real': npy(lr_), 'gan.fake': npy(lf_)})
    # LPIPS (random, seeded backbone + random lin layers; weights are regenerated from the seed by the test)
    torch.manual_seed(77)
    lin_w = {f'{i}.1.weight': torch.rand(1, c, 1, 1, generator=g) for i, c in enumerate([64, 128, 256, 512, 512])}
    ref_lpips.get_state_dict = lambda *a, **k: lin_w
    lp = ref_lpips.LPIPS('vgg')
    imgs = torch.rand(2, 3, 32, 32, generator=g) * 2 - 1
    rec = (imgs + 0.3 * torch.randn(2, 3, 32, 32, generator=g)).clamp(-1, 1).requires_grad_(True)
    val = lp(imgs, rec)
    out.update({'lpips.images': npy(imgs), 'lpips.recon': npy(rec), 'lpips.value': npy(val),
                'lpips.drecon': npy(torch.autograd.grad(val, rec)[0])})
    out.update({f'lpips.lin{i}': npy(v).reshape(-1) for i, v in enumerate(lin_w.values())})
    # recon loss terms
    l1 = (imgs - rec).abs().mean()
    l2 = (imgs - rec).pow(2).mean()
    gl1, = torch.autograd.grad(0.8 * l1 + 0.2 * l2, rec)
    out.update({'recon.l1': npy(l1), 'recon.l2': npy(l2), 'recon.d': npy(gl1)})
    save('gan', **out)


if __name__ == '__main__' and 'gan' in sys.argv[1:]:
    gen_gan()

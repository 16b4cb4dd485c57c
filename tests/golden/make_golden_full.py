"""Full-size golden vectors: every BASELINE.json config exercised at its real quantizer / discriminator size by the
REFERENCE's own modules on CPU (build container only; the reference never travels):

    python tests/golden/make_golden_full.py [ema] [entropy] [gumbel] [disc] [config1]

Inputs and weights are regenerated from seeds by ``tests/golden/seeded.py`` on both sides; the ``full_*.npz`` files
hold what the reference computed from them -- indices exactly, small tensors in full, large tensors as strided rows
plus a six-number summary (sum, norm, four random projections).  Only data is written; no reference source.
"""
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

REF = os.environ.get('VQK_REFERENCE', '/root/reference')
sys.path.insert(0, REF)
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import seeded as S  # noqa: E402
from vqvae.modules.autoencoder import Encoder, Decoder  # noqa: E402
from vqvae.modules.vector_quantizers import (VectorQuantizer, EMAVectorQuantizer, EntropyVectorQuantizer,  # noqa: E402
                                             GumbelVectorQuantizer)

torch.set_num_threads(8)


def npy(t):
    return t.detach().cpu().numpy().copy()


def save(name, **arrs):
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **arrs)
    print(f'{name}: {os.path.getsize(path) / 1024:.1f} KiB, {len(arrs)} arrays', flush=True)


def gen_ema():
    """EMAVectorQuantizer.forward in training mode at K=1024, D=256, N=8192 (vector_quantizers.py:128-180)"""
    i = S.ema_full_inputs()
    q = EMAVectorQuantizer(1024, 256, i['beta'], i['decay'], i['eps'])
    with torch.no_grad():
        q.codebook.weight.copy_(i['e'])
        q.ema_count.copy_(i['ema_count'])
        q.ema_weight.copy_(i['ema_weight'])
    q.train()
    z = i['z'].clone().requires_grad_(True)
    qz, idx, loss = q(z)
    dz, = torch.autograd.grad([qz, loss], [z], [i['dq'], torch.tensor(1.0)])
    assert idx.max() < 32768
    save('full_ema', idx=npy(idx).astype(np.int16), loss=npy(loss), q_sum=S.summary(qz, 'ema.q'),
         dz_sum=S.summary(dz, 'ema.dz'), count_after=npy(q.ema_count),
         weight_after_sum=S.summary(q.ema_weight, 'ema.weight'), weight_after_rows=npy(q.ema_weight[::16]),
         cb_after_sum=S.summary(q.codebook.weight, 'ema.cb'), cb_after_rows=npy(q.codebook.weight[::16]),
         used_codes=np.int64(len(torch.unique(idx))))


def gen_entropy():
    """EntropyVectorQuantizer fwd+bwd at K=8192, D=256, N=4096 (vector_quantizers.py:290-356), the config's T=0.01 and
    a soft T=1.0 (where the entropy gradient is not vanishing)"""
    out = {}
    for tag, temp in (('t001', 0.01), ('t1', 1.0)):
        i = S.entropy_full_inputs(temp)
        q = EntropyVectorQuantizer(8192, 256, i['ratio'], temp, 'softmax', i['beta'])
        with torch.no_grad():
            q.codebook.weight.copy_(i['e'])
        z = i['z'].clone().requires_grad_(True)
        t0 = time.time()
        qz, idx, loss = q(z)
        dz, de = torch.autograd.grad([qz, loss], [z, q.codebook.weight], [i['dq'], torch.tensor(1.0)])
        print(f'entropy {tag}: {time.time() - t0:.1f} s, loss {loss.item():.6f}, codes used {len(torch.unique(idx))}', flush=True)
        out.update({f'{tag}.idx': npy(idx).astype(np.int16), f'{tag}.loss': npy(loss),
                    f'{tag}.dz_sum': S.summary(dz, f'ent.{tag}.dz'), f'{tag}.dz_rows': npy(dz[:, :, ::8, ::8]),
                    f'{tag}.de_sum': S.summary(de, f'ent.{tag}.de'), f'{tag}.de_rows': npy(de[::64])})
    save('full_entropy', **out)


def gen_gumbel():
    """GumbelVectorQuantizer fwd+bwd at K=1024, D=256, N=4096 with injected noise (vector_quantizers.py:223-245)"""
    i = S.gumbel_full_inputs()
    q = GumbelVectorQuantizer(1024, 256, False, i['tau'], i['kl_cost'])
    with torch.no_grad():
        q.codebook.weight.copy_(i['e'])
        q.x_to_logits.weight.copy_(i['w'])
        q.x_to_logits.bias.copy_(i['b'])
    q.train()
    x = i['x'].clone().requires_grad_(True)
    # F.gumbel_softmax draws torch.empty_like(logits).exponential_() first: serve the injected tensor
    orig = torch.Tensor.exponential_

    def fake_exponential_(self, *a, **k):
        return self.copy_(i['noise'])
    torch.Tensor.exponential_ = fake_exponential_
    try:
        qz, idx, loss = q(x)
    finally:
        torch.Tensor.exponential_ = orig
    gr = torch.autograd.grad([qz, loss], [x, q.codebook.weight, q.x_to_logits.weight, q.x_to_logits.bias],
                             [i['dq'], torch.tensor(1.0)])
    print(f'gumbel: loss {loss.item():.6e}, codes used {len(torch.unique(idx))}', flush=True)
    save('full_gumbel', idx=npy(idx).astype(np.int16), loss=npy(loss), q_sum=S.summary(qz, 'gum.q'),
         q_rows=npy(qz[:, :, ::8, ::8]), dx_sum=S.summary(gr[0], 'gum.dx'), dx_rows=npy(gr[0][::8, ::64, ::4, ::4]),
         de_sum=S.summary(gr[1], 'gum.de'), de_rows=npy(gr[1][::32]), dw_sum=S.summary(gr[2], 'gum.dw'),
         dw_rows=npy(gr[2][::64, ::16]), db=npy(gr[3]))


def gen_disc():
    """Discriminator(256) on [4,3,256,256]: logits, all first-order gradients, R1 value + its parameter gradients
    (discriminator.py:360-412, loss.py:98-112)"""
    import types
    tv = types.ModuleType('torchvision')
    tv.models = types.ModuleType('torchvision.models')
    sys.modules.setdefault('torchvision', tv)
    sys.modules.setdefault('torchvision.models', tv.models)
    from vqvae.modules.loss.stylegan2_discriminator.discriminator import Discriminator
    i = S.disc256_inputs()
    d = Discriminator(256)
    S.fill_named(list(d.named_parameters()), i['seed'], 'discriminator')
    named = list(d.named_parameters())
    x = i['x'].clone().requires_grad_(True)
    t0 = time.time()
    logits = d(x)
    grads = torch.autograd.grad((logits * i['r']).sum(), [x] + [p for _, p in named])
    print(f'disc fwd+bwd {time.time() - t0:.1f} s, logits {logits.view(-1).tolist()}', flush=True)
    out = dict(logits=npy(logits), dx_sum=S.summary(grads[0], 'd256.dx'), dx_rows=npy(grads[0][:, :, ::16, ::16]))
    out.update({'g.' + n: S.summary(gr, 'd256.g.' + n) for (n, _), gr in zip(named, grads[1:])})
    xr = i['x'].clone().requires_grad_(True)
    t0 = time.time()
    lg = d(xr)
    gimg, = torch.autograd.grad(lg.sum(), xr, create_graph=True)
    r1 = 10.0 * gimg.pow(2).view(gimg.shape[0], -1).sum(1).mean()
    r1g = torch.autograd.grad(r1, [p for _, p in named], allow_unused=True)
    print(f'disc R1 {time.time() - t0:.1f} s, r1 {r1.item():.6e}', flush=True)
    out.update({'r1.value': npy(r1), 'r1.gimg_sum': S.summary(gimg, 'd256.gimg')})
    out.update({'r1g.' + n: S.summary(gr, 'd256.r1g.' + n) for (n, _), gr in zip(named, r1g) if gr is not None})
    out['n_params'] = np.int64(sum(p.numel() for _, p in named))
    save('full_disc256', **out)


def gen_config1():
    """BASELINE config 1: standard_vqvae.yaml architecture at 64x64, bs=8: forward, losses, all 144 parameter gradients
    and the parameters after one torch.optim.AdamW step with the reference's two groups (model.py:419-428)"""
    i = S.config1_inputs()
    enc = Encoder(128, 2, (1, 2, 2, 4), 256)
    dec = Decoder(128, 2, (1, 2, 2, 4), 256)
    q = VectorQuantizer(1024, 256, 0.25)
    named = ([('encoder.' + n, p) for n, p in enc.named_parameters()]
             + [('decoder.' + n, p) for n, p in dec.named_parameters()]
             + [('quantizer.' + n, p) for n, p in q.named_parameters()])
    S.fill_named(named, i['seed'])
    with torch.no_grad():
        q.codebook.weight.copy_(i['codebook'])
    x = (i['images'].clamp(0, 1) - 0.5) / 0.5
    z = enc(x)
    print('z rms', z.pow(2).mean().sqrt().item(), flush=True)
    qz, idx, ql = q(z)
    recon = dec(qz)
    l2 = F.mse_loss(recon, x)
    loss = ql + l2
    grads = torch.autograd.grad(loss, [p for _, p in named])
    print(f'config1: loss {loss.item():.6f}, q_loss {ql.item():.6f}, codes used {len(torch.unique(idx))}', flush=True)
    out = dict(idx=npy(idx).astype(np.int16), z_sum=S.summary(z, 'c1.z'), recon_sum=S.summary(recon, 'c1.recon'),
               recon_rows=npy(recon[:, :, ::8, ::8]), q_loss=npy(ql), l2=npy(l2), loss=npy(loss))
    out.update({'g.' + n: S.summary(gr, 'c1.g.' + n) for (n, _), gr in zip(named, grads)})
    is_decay = lambda n, p: p.ndim == 4 and p.shape[0] != 1 and not n.rsplit('.', 2)[-2].startswith('norm')
    decay = [p for n, p in named if is_decay(n, p)]
    no_decay = [p for n, p in named if not is_decay(n, p)]
    opt = torch.optim.AdamW([{'params': decay, 'weight_decay': 1e-4}, {'params': no_decay, 'weight_decay': 0.0}],
                            lr=1e-4, betas=(0.0, 0.99), eps=1e-8, weight_decay=1e-4)
    for (n, p), gr in zip(named, grads):
        p.grad = gr
    opt.step()
    out.update({'p.' + n: S.summary(p, 'c1.p.' + n) for n, p in named})
    out['n_tensors'] = np.int64(len(named))
    save('full_config1', **out)


def gen_utils():
    """BaseVectorQuantizer utilities (base_quantizer.py:54-102): usage statistics, dead-code re-initialisation under a fixed
    CPU seed, codes_to_vec"""
    g = torch.Generator().manual_seed(808)
    k, d = 64, 16
    count = torch.randint(0, 40, (k,), generator=g).float()
    count[torch.randperm(k, generator=g)[:20]] = 0.0
    q = VectorQuantizer(k, d, 0.25)
    with torch.no_grad():
        q.codebook.weight.copy_(torch.randn(k, d, generator=g))
    cb0 = npy(q.codebook.weight)
    p, perplexity, used = q.get_codebook_usage(count)
    codes = torch.randint(0, k, (3, 7), generator=g)
    vec = q.codes_to_vec(codes)
    torch.manual_seed(4242)
    q.reinit_unused_codes(p)
    save('quantizer_utils', count=npy(count), p=npy(p), perplexity=np.float64(perplexity), used=np.float64(used),
         cb0=cb0, cb1=npy(q.codebook.weight), codes=npy(codes), vec=npy(vec), reinit_seed=np.int64(4242))


if __name__ == '__main__':
    which = sys.argv[1:] or ['ema', 'entropy', 'gumbel', 'disc', 'config1', 'utils']
    for w in which:
        {'ema': gen_ema, 'entropy': gen_entropy, 'gumbel': gen_gumbel, 'disc': gen_disc, 'config1': gen_config1,
         'utils': gen_utils}[w]()

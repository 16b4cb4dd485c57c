"""GPU parity of the whole train step (LightningModule surface -> HIP kernels) against the golden vectors the
reference produced (tests/golden/train_step_*.npz) and against the CPU oracle."""
import importlib

import numpy as np
import pytest
import torch

from oracle import vqvae_oracle as O

pytestmark = pytest.mark.gpu

model_mod = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd.model')
trainer_mod = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd.trainer')
DEV = 'cuda:0'
T = torch.from_numpy

AE = dict(channels=32, num_res_blocks=1, channel_multipliers=(1, 2))
TC = dict(lr=1e-4, betas=(0.0, 0.99), eps=1e-8, weight_decay=1e-4, warmup_epochs=None, decay_epochs=None)
QP = {'standard': dict(commitment_cost=0.25), 'ema': dict(commitment_cost=0.25, decay=0.95, epsilon=1e-5),
      'entropy': dict(commitment_cost=0.25, ent_loss_ratio=0.1, ent_temperature=0.01, ent_loss_type='softmax')}


def rel(a, b, floor=1e-7):
    """||a-b|| / (||b|| + floor*sqrt(n)): gradients that are analytically zero (a bias in front of a
    one-channel-per-group GroupNorm) are pure rounding noise and must not fail a relative test."""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / (b.norm() + floor * b.numel() ** 0.5)).item()


# The last Upsample's conv bias feeds a GroupNorm with ONE channel per group (32 channels / 32 groups): a
# per-channel constant is removed by the mean subtraction, so its gradient is analytically zero and both the
# reference's and our values (~1e-9) are rounding noise.
ZERO_GRAD = {'decoder.blocks.3.conv.bias'}


def build(golden, qtype, dtype=torch.float32):
    base = golden('train_step_standard')
    g = golden(f'train_step_{qtype}')
    qc = dict(num_embeddings=64, embedding_dim=16, reinit_every_n_epochs=None, type=qtype, params=QP[qtype])
    m = model_mod.VQVAE(32, AE, qc, None, TC, compute_dtype=dtype)
    sd = {k: T(v) for k, v in base.items() if k.startswith(('encoder.', 'decoder.'))}
    sd.update({k: T(v) for k, v in (base if qtype == 'standard' else g).items() if k.startswith('quantizer.')})
    missing = m.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return base, g, m.to(DEV).train()


@pytest.mark.parametrize('products', [torch.float32, 'bf16x3'])
@pytest.mark.parametrize('qtype', ['standard', 'ema', 'entropy'])
def test_train_step_fp32_golden(golden, qtype, products):
    """products: torch.float32 = exact fp32 MFMA products (the reference mode); 'bf16x3' = the parity-grade mode on the bf16 matrix
    pipe (three bf16 products per multiply-add, csrc/conv_x3.hip) -- SAME tolerances"""
    base, g, m = build(golden, qtype, products)
    tr = trainer_mod.MiniTrainer(num_training_batches=1)
    opt = tr.attach(m)[0]
    images = T(base['images']).to(DEV)
    recon, e_loss, idx = m(m.preprocess_batch(images))
    # forward() re-runs the EMA update; rebuild so the step below starts from the golden state
    assert np.array_equal(idx.cpu().numpy(), g['out.idx'])                      # codebook indices: bit-exact
    np.testing.assert_allclose(recon.detach().float().cpu().numpy(), g["out.recon"], rtol=1e-3, atol=2e-5)
    np.testing.assert_allclose(e_loss.item(), g['out.q_loss'], rtol=1e-4)

    base, g, m = build(golden, qtype, products)
    opt = tr.attach(m)[0]
    opt.zero_grad()
    loss = m.training_step(images, 0)
    loss.backward()
    np.testing.assert_allclose(loss.item(), g['out.loss'], rtol=1e-4)
    named = dict(m.named_parameters())
    checked = 0
    for k, v in g.items():
        if k.startswith('grad.') and k[5:] not in ZERO_GRAD:
            assert rel(named[k[5:]].grad, T(v)) < 1e-3, k
            checked += 1
    assert checked >= 10
    if qtype == 'ema':
        np.testing.assert_allclose(m.quantizer.ema_count.cpu().numpy(), g['after.ema_count'], rtol=1e-5, atol=1e-7)
        # ema_weight = decay * old + (1 - decay) * (sum of the latents of a code): it inherits the latents' own error -- fp32 rounding in
        # the exact mode, ~2^-17 per conv product (1e-5 of the latents' magnitude, measured 1.7e-6 absolute) with split products
        tol = dict(rtol=1e-5, atol=1e-7) if products == torch.float32 else dict(rtol=1e-4, atol=5e-6)
        np.testing.assert_allclose(m.quantizer.ema_weight.cpu().numpy(), g['after.ema_weight'], **tol)
        np.testing.assert_allclose(m.quantizer.codebook.weight.detach().cpu().numpy(), g['after.codebook.weight'],
                                   rtol=1e-4, atol=1e-6 if products == torch.float32 else 2e-5)      # (= ema_weight / count: same inheritance)
    if qtype == 'standard':                       # one AdamW step with the reference's two groups
        opt.step()
        decay = {n for n, _ in m.optimizer_groups()[0]}
        assert decay == set(g['decay_names'].tolist())
        # beta1 = 0: the first AdamW step is lr * g / (|g| + eps'), i.e. lr * sign(g) unless |g| ~ 1e-8, where a
        # 1e-9 difference in g moves the step by several percent.  So: every element within 2*lr of the golden
        # value, and all but a vanishing fraction within the tight tolerance.
        for k, v in g.items():
            if k.startswith('stepped.') and k[8:] not in ZERO_GRAD:
                got = named[k[8:]].detach().cpu().numpy()
                assert np.abs(got - v).max() <= 2.1e-4, k
                bad = ~np.isclose(got, v, rtol=1e-5, atol=2e-7)
                assert bad.mean() <= 1e-3, (k, bad.mean())


def test_bf16_mode_tracks_fp32_mode(golden):
    """throughput mode (bf16 storage, bf16 MFMA, fp32 accumulate) against the parity mode ON THE SAME GPU, with the
    quantizer taken out of the comparison (a flipped near-tie index is a discontinuity, not rounding noise):
    encoder latents, decoder output and parameter gradients of a fixed linear functional."""
    base, g, m32 = build(golden, 'standard', torch.float32)
    _, _, m16 = build(golden, 'standard', torch.bfloat16)
    x = m32.preprocess_batch(T(base['images']).to(DEV))
    gen = torch.Generator().manual_seed(5)
    probe = torch.randn(4, 3, 32, 32, generator=gen).to(DEV)
    outs = {}
    for tag, m in (('f32', m32), ('bf16', m16)):
        z = m.encoder(x)
        recon = m.decoder(z)                       # decoder fed with the (continuous) latents
        loss = (recon.float() * probe).sum()
        plist = [p for n, p in m.named_parameters() if n.startswith(('encoder.', 'decoder.')) and n not in ZERO_GRAD]
        grads = torch.autograd.grad(loss, plist)
        outs[tag] = (z.detach(), recon.detach().float(), grads)
    assert rel(outs['bf16'][0], outs['f32'][0]) < 3e-2
    assert rel(outs['bf16'][1], outs['f32'][1]) < 3e-2
    errs = [rel(a, b, floor=1e-5) for a, b in zip(outs['bf16'][2], outs['f32'][2])]
    assert np.median(errs) < 5e-2 and max(errs) < 0.3, (np.median(errs), max(errs))
    # and the full bf16 step runs and lands near the fp32 loss
    tr = trainer_mod.MiniTrainer(num_training_batches=1)
    opt = tr.attach(m16)[0]
    opt.zero_grad()
    loss = m16.training_step(T(base['images']).to(DEV), 0)
    loss.backward()
    opt.step()
    assert abs(loss.item() - float(g['out.loss'])) / float(g['out.loss']) < 5e-2


def test_reference_param_set_reproduces_collision(golden):
    """optimizer_param_set='reference': encoder tensors whose relative name also exists in the decoder are dropped."""
    qc = dict(num_embeddings=64, embedding_dim=16, reinit_every_n_epochs=None, type='standard', params=QP['standard'])
    m_all = model_mod.VQVAE(32, AE, qc, None, TC)
    m_ref = model_mod.VQVAE(32, AE, qc, None, TC, optimizer_param_set='reference')
    n_all = sum(len(x) for x in m_all.optimizer_groups())
    n_ref = sum(len(x) for x in m_ref.optimizer_groups())
    assert n_all == len([p for p in m_all.parameters() if p.requires_grad])
    assert n_ref < n_all
    kept = {n for grp in m_ref.optimizer_groups() for n, _ in grp}
    assert 'encoder.conv_in.weight' not in kept and 'decoder.conv_in.weight' in kept


def test_multi_step_loss_decreases():
    torch.manual_seed(0)
    qc = dict(num_embeddings=64, embedding_dim=16, reinit_every_n_epochs=None, type='standard', params=QP['standard'])
    tc = dict(TC, lr=2e-3)
    m = model_mod.VQVAE(32, AE, qc, None, tc).to(DEV)
    images = torch.rand(8, 3, 32, 32, device=DEV)
    tr = trainer_mod.MiniTrainer(max_epochs=1)
    first = None
    tr.attach(m)
    m.train()
    m.on_train_start()
    for i in range(30):
        loss = tr.train_batch(m, images, i)
        first = first if first is not None else loss.item()
    assert loss.item() < first


@pytest.mark.parametrize('qtype', ['standard', 'ema'])
def test_graph_replay_matches_eager(qtype):
    """hipGraph replay of fwd+bwd (MiniTrainer.capture) walks the same trajectory as eager launches.  EMA: the graph path
    defers the statistics all-reduce + update kernel to after the replay (EMAVectorQuantizer.finish_update) -- the
    codebook / EMA buffers must follow the inline update of the eager path."""
    qc = dict(num_embeddings=64, embedding_dim=16, reinit_every_n_epochs=None, type=qtype, params=QP[qtype])
    tc = dict(TC, lr=1e-3)
    images = torch.rand(4, 3, 32, 32, generator=torch.Generator().manual_seed(3)).to(DEV)
    losses, state = {}, {}
    for mode in ('eager', 'graph'):
        torch.manual_seed(0)
        m = model_mod.VQVAE(32, AE, qc, None, tc).to(DEV).train()
        tr = trainer_mod.MiniTrainer(num_training_batches=100)
        tr.attach(m)
        m.on_train_start()
        out = []
        if mode == 'graph':
            tr.capture(m, images, warmup=2)
            assert getattr(m.quantizer, 'defer_update', True)
            for i in range(4):
                out.append(tr.train_batch_graphed(m, images, 2 + i).item())
        else:
            for i in range(6):
                out.append(tr.train_batch(m, images, i).item())
            out = out[2:]
        losses[mode] = out
        state[mode] = {k: v.detach().float().cpu().clone() for k, v in m.quantizer.state_dict().items()}
    np.testing.assert_allclose(losses['graph'], losses['eager'], rtol=2e-3)
    if qtype == 'ema':
        for k in ('ema_count', 'ema_weight', 'codebook.weight'):
            np.testing.assert_allclose(state['graph'][k].numpy(), state['eager'][k].numpy(), rtol=2e-3, atol=1e-5)

"""LightningModule-level golden vectors captured from the REFERENCE's real ``vqvae.model.VQVAE`` (build container only):

    python tests/golden/make_golden_model.py [optim] [gan]

``vqvae.model`` imports packages that are not installed here (pytorch_lightning, scheduling_utils, torchvision, wandb,
torchmetrics, kornia).  They are replaced by minimal ``sys.modules`` stand-ins: a LightningModule that is
an ``nn.Module`` with ``log`` / ``optimizers`` / ``manual_backward``; identity augmentation; ``Normalize`` /
``Denormalize`` as (x-m)/s and x*s+m; a hand-assembled cfg-D ``vgg16().features``.  None of the stand-ins carries
arithmetic of the path except the trivial normalisation.  Only data is written (names, scalars, summaries).

  optim -> model_optim.npz: the tensors ``VQVAE.configure_optimizers`` really hands to AdamW (the reference's
           name-collision: 91 of 144 for standard_vqvae.yaml), in group order, + the same for gumbel_vqgan.yaml
  gan   -> model_gan_step.npz: one real ``training_step`` of gumbel_vqgan.yaml (start_epoch 0, R1 every step,
           injected Gumbel noise, seeded weights) at 64x64, bs=4, for use_adaptive in {False, True}: the nine logged
           scalars, g_weight, r1, every .grad left behind and every parameter after the two optimizer steps
"""
import os
import sys
import types

import numpy as np
import torch
import yaml

REF = os.environ.get('VQK_REFERENCE', '/root/reference')
sys.path.insert(0, REF)
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import seeded as S  # noqa: E402

torch.set_num_threads(8)


def install_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class LightningModule(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.trainer = None
            self.current_epoch = 0
            self.automatic_optimization = True
            self.logged = {}

        def log(self, name, value, **_):
            self.logged[name] = value

        def optimizers(self):
            return self.trainer.optimizers

        def manual_backward(self, loss):
            loss.backward()

    mod('pytorch_lightning', LightningModule=LightningModule)

    class _Sched:
        def __init__(self, *a):
            self.a = a

        def step(self, i):
            raise RuntimeError('schedule stand-in: not used by the captured cases')

        def destroy(self):
            pass
    mod('scheduling_utils')
    mod('scheduling_utils.schedulers_cpp', LinearScheduler=_Sched, CosineScheduler=_Sched, LinearCosineScheduler=_Sched)

    def vgg16(weights=None):
        cfg = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512, 'M']
        layers, cin = [], 3
        for v in cfg:
            if v == 'M':
                layers.append(torch.nn.MaxPool2d(2, 2))
            else:
                layers += [torch.nn.Conv2d(cin, v, 3, padding=1), torch.nn.ReLU(inplace=True)]
                cin = v
        return types.SimpleNamespace(features=torch.nn.Sequential(*layers))
    models = mod('torchvision.models', vgg16=vgg16, VGG16_Weights=types.SimpleNamespace(DEFAULT=None))
    tvu = mod('torchvision.utils', make_grid=lambda *a, **k: None)
    tvt = mod('torchvision.transforms', ConvertImageDtype=lambda *a, **k: None)
    mod('torchvision', models=models, utils=tvu, transforms=tvt)
    mod('wandb', Image=lambda *a, **k: None)

    class _Metric:
        def __init__(self, *a, **k):
            pass
    mod('torchmetrics', MeanSquaredError=_Metric)
    mod('torchmetrics.image')
    mod('torchmetrics.image.fid', FrechetInceptionDistance=_Metric)
    mod('torchmetrics.image.ssim', StructuralSimilarityIndexMeasure=_Metric)
    mod('torchmetrics.image.psnr', PeakSignalNoiseRatio=_Metric)

    class _Identity:
        def __init__(self, *a, **k):
            pass

        def __call__(self, x):
            return x

    class Normalize:
        def __init__(self, mean, std):
            self.m, self.s = mean.view(1, -1, 1, 1), std.view(1, -1, 1, 1)

        def __call__(self, x):
            return (x - self.m) / self.s

    class Denormalize(Normalize):
        def __call__(self, x):
            return x * self.s + self.m
    mod('kornia')
    mod('kornia.augmentation', AugmentationSequential=_Identity, RandomResizedCrop=_Identity,
        RandomHorizontalFlip=_Identity, Normalize=Normalize, Denormalize=Denormalize)


def load_conf(name):
    return yaml.safe_load(open(os.path.join(REF, 'example_confs', name)))


def t_conf_of(conf, lr=None):
    tr = conf['training']
    return dict(lr=float(tr['base_lr']) if lr is None else lr, betas=tr['betas'], eps=tr['eps'],
                weight_decay=tr['weight_decay'], warmup_epochs=tr.get('warmup_epochs'), decay_epochs=tr.get('decay_epochs'))


def save(name, **arrs):
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **arrs)
    print(f'{name}: {os.path.getsize(path) / 1024:.1f} KiB, {len(arrs)} arrays', flush=True)


def patch_lpips_download():
    """the LPIPS lin weights are downloaded by the reference (utils.py:13-20); offline: placeholder ones, overwritten by seed"""
    from vqvae.modules.loss.lpips_pytorch.modules import lpips as ref_lpips
    ref_lpips.get_state_dict = lambda *a, **k: {f'{i}.1.weight': torch.ones(1, c, 1, 1)
                                                for i, c in enumerate([64, 128, 256, 512, 512])}


def gen_optim():
    from vqvae.model import VQVAE
    patch_lpips_download()
    out = {}
    for tag, yml, size in (('standard', 'standard_vqvae.yaml', 256), ('ema', 'ema_vqvae.yaml', 256),
                           ('gumbel_vqgan', 'gumbel_vqgan.yaml', 32)):
        conf = load_conf(yml)
        m = VQVAE(size, conf['autoencoder'], conf['quantizer'], conf.get('loss'), t_conf_of(conf))
        names = {id(p): n for n, p in m.named_parameters()}
        opt = m.configure_optimizers()
        ae = opt[0][0] if isinstance(opt, tuple) else opt
        for gi, g in enumerate(ae.param_groups):
            out[f'{tag}.group{gi}'] = np.array([names[id(p)] for p in g['params']])
            out[f'{tag}.group{gi}.weight_decay'] = np.float64(g['weight_decay'])
        out[f'{tag}.n_trainable'] = np.int64(sum(1 for _, p in m.named_parameters()
                                               if p.requires_grad and not _n_is_criterion(names[id(p)])))
        if isinstance(opt, tuple):
            out[f'{tag}.disc'] = np.array([names[id(p)] for g in opt[0][1].param_groups for p in g['params']])
        print(tag, [len(g['params']) for g in ae.param_groups], flush=True)
    save('model_optim', **out)


def _n_is_criterion(n):
    return n.startswith('criterion.')


def gen_gan():
    from vqvae.model import VQVAE
    patch_lpips_download()
    conf = load_conf('gumbel_vqgan.yaml')
    size, bs, seed = 64, 4, 7007
    out = {}
    g = torch.Generator().manual_seed(seed)
    images = torch.rand(bs, 3, size, size, generator=g)
    noise = torch.empty(bs, conf['quantizer']['num_embeddings'], size // 16, size // 16).exponential_(generator=g)
    out['images'] = images.numpy()
    for tag, adaptive, gw in (('fixed', False, 0.1), ('adaptive', True, 0.8)):
        l_conf = dict(conf['loss'])
        l_conf['adversarial_params'] = dict(l_conf['adversarial_params'], start_epoch=0, r1_reg_every=1,
                                            use_adaptive=adaptive, g_weight=gw)
        q_conf = dict(conf['quantizer'])
        m = VQVAE(size, conf['autoencoder'], q_conf, l_conf, t_conf_of(conf, lr=1e-4))
        S.fill_vqgan(m, seed)
        m.train()
        opts, _ = m.configure_optimizers()
        m.trainer = types.SimpleNamespace(num_training_batches=10, optimizers=opts)
        before = {n: p.detach().clone() for n, p in m.named_parameters()}
        orig = torch.Tensor.exponential_
        torch.Tensor.exponential_ = lambda self, *a, **k: self.copy_(noise)
        try:
            m.training_step(images, 0)
        finally:
            torch.Tensor.exponential_ = orig
        for k, v in m.logged.items():
            out[f'{tag}.log.{k}'] = np.float64(float(v))
        names = {id(p): n for n, p in m.named_parameters()}
        out[f'{tag}.ae_opt'] = np.array([names[id(p)] for gr in opts[0].param_groups for p in gr['params']])
        for n, p in m.named_parameters():
            if p.grad is not None:
                out[f'{tag}.grad.{n}'] = S.summary(p.grad, f'gan.{tag}.grad.{n}')
            if not torch.equal(p.detach(), before[n]):
                out[f'{tag}.after.{n}'] = S.summary(p, f'gan.{tag}.after.{n}')
        print(tag, {k: float(v) for k, v in m.logged.items()}, flush=True)
    save('model_gan_step', **out)


if __name__ == '__main__':
    install_stubs()
    which = sys.argv[1:] or ['optim', 'gan']
    for w in which:
        {'optim': gen_optim, 'gan': gen_gan}[w]()

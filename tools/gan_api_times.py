#!/usr/bin/env python
"""Where one eager VQ-GAN step (config 4, bs 16) spends its device time, per C-ABI entry point: HIP events around EVERY
libvqk call, summed by function (and by the phase of the step: AE half / discriminator half)."""
import collections
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ops = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd.ops')
native = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd._native')
model_mod = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd.model')
trainer_mod = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd.trainer')
train_mod = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd.train')
lib = native.lib()
log = []
phase = ['ae']
SKIP = {'vqk_set_tuning', 'vqk_reset_tuning', 'vqk_tuning_count', 'vqk_tuning_name', 'vqk_arch', 'vqk_status_str',
        'vqk_conv_weight_layout', 'vqk_conv_packed_elems', 'vqk_conv2d_s2_supported'}


def wrap(name):
    orig = getattr(lib, name)

    def f(*a):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = orig(*a); e1.record()
        log.append((phase[0], name, e0, e1))
        return r
    setattr(lib, name, f)


for nm in native.EXPORTS:
    if nm not in SKIP and 'ws_bytes' not in nm and 'set_' not in nm:
        wrap(nm)

dev = torch.device('cuda:0')
torch.manual_seed(1234)
conf = train_mod.get_model_conf(os.path.join(ROOT, 'example_confs', 'gumbel_vqgan.yaml'))
run = train_mod.derive_run_config(conf, 1, {'training.cumulative_bs': 16, 'loss.adversarial_params.start_epoch': 0})
m = model_mod.VQVAE(run['image_size'], run['ae_conf'], run['q_conf'], run['l_conf'], run['t_conf'], compute_dtype=torch.bfloat16).to(dev)
m.criterion.discriminator.compute_dtype = torch.bfloat16
m.criterion.perceptual_loss.net.compute_dtype = torch.bfloat16
m.train()
tr = trainer_mod.MiniTrainer(num_training_batches=10)
tr.attach(m)
m.on_train_start()
x = torch.rand(16, 3, 256, 256).to(dev)
for i in range(1, 4):
    tr.train_batch(m, x, i)
torch.cuda.synchronize()
log.clear()
orig_d = m._gan_disc_half


def disc_half(step):
    phase[0] = 'disc'
    try:
        return orig_d(step)
    finally:
        phase[0] = 'ae'


m._gan_disc_half = disc_half
t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0.record()
tr.train_batch(m, x, 5)                  # no R1 on this step
t1.record()
torch.cuda.synchronize()
agg = collections.OrderedDict()
for ph, name, e0, e1 in log:
    c, t = agg.get((ph, name), (0, 0.0))
    agg[(ph, name)] = (c + 1, t + e0.elapsed_time(e1) * 1e3)
for ph in ('ae', 'disc'):
    tot = sum(t for (p, _), (c, t) in agg.items() if p == ph)
    print(f'--- {ph} half: {tot / 1e3:.2f} ms inside libvqk calls')
    for (p, name), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        if p == ph and t > 40:
            print(f'{t:9.1f} us  x{c:<4d} {name}')
print(f'eager step wall (events): {t0.elapsed_time(t1):.2f} ms')

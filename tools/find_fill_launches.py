#!/usr/bin/env python
"""Which host lines issue the NON-vqk launches of a train step (ATen fills / copies / elementwise kernels, hipMemcpy)?
torch.profiler with stacks over one eager step; printed: count, op, innermost frame inside this repository.
Usage: python tools/find_fill_launches.py [--gan] [--batch 8]"""
import argparse
import collections
import importlib
import os
import sys

import torch

R = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--gan', action='store_true')
ap.add_argument('--batch', type=int, default=8)
ap.add_argument('--quantizer', default='standard')
a = ap.parse_args()
trainer_mod = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd.trainer')
model_mod = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd.model')
ns = argparse.Namespace(config=None, quantizer='gumbel' if a.gan else a.quantizer, batch=a.batch, image_size=256, codebook=None, gan=a.gan)
run, _, _ = bench.run_config(ns, 1)
torch.manual_seed(0)
m = model_mod.VQVAE(256, run['ae_conf'], run['q_conf'], run['l_conf'], run['t_conf'], compute_dtype=torch.bfloat16).cuda().train()
if a.gan:
    m.criterion.discriminator.compute_dtype = torch.bfloat16
    m.criterion.perceptual_loss.net.compute_dtype = torch.bfloat16
tr = trainer_mod.MiniTrainer(num_training_batches=100)
tr.attach(m)
m.on_train_start()
x = torch.rand(a.batch, 3, 256, 256).cuda()
for i in range(3):
    tr.train_batch(m, x, i)
torch.cuda.synchronize()
import traceback  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

NOLAUNCH = ('empty', 'view', 'reshape', 'permute', 'as_strided', 'detach', 'alias', 'expand', 'slice', 'select', 'unsqueeze', 'squeeze',
            't.', 'transpose', '_unsafe_view', 'size', 'stride', 'is_', 'storage_offset', 'numel', 'dim', 'sym_', 'item', '_local_scalar',
            'lift_fresh', 'record_stream', 'resize_', 'set_', 'contiguous', 'chunk', 'split', 'unbind', 'narrow', 'result_type', '_has_',
            'is_pinned', 'prim', 'empty_like', 'empty_strided', 'new_empty', 'unfold', 'flatten', '_to_copy')


class Spy(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.cnt = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(k in name for k in NOLAUNCH) or '_to_copy' in name:
            cuda = any(torch.is_tensor(x) and x.is_cuda for x in list(args) + list((kwargs or {}).values()))
            if cuda or 'zeros' in name or 'ones' in name or 'full' in name or 'arange' in name or 'tensor' in name:
                fr = [f for f in traceback.extract_stack() if 'vqvae-vqgan' in f.filename and 'find_fill' not in f.filename]
                where = f'{os.path.basename(fr[-1].filename)}:{fr[-1].lineno} {fr[-1].name}' if fr else '?'
                up = f' <- {os.path.basename(fr[-2].filename)}:{fr[-2].lineno}' if len(fr) > 1 else ''
                self.cnt[(name, where + up)] += 1
        return func(*args, **(kwargs or {}))


torch.autograd.set_multithreading_enabled(False)       # the backward on THIS thread: the dispatch mode is thread-local
spy = Spy()
with spy:
    tr.train_batch(m, x, 3)            # (VQ-GAN: step 3 -- no R1 term with r1_reg_every = 16)
    torch.cuda.synchronize()
tot = 0
for (n, st), c in spy.cnt.most_common(120):
    tot += c
    print(f'{c:4d} {n:34s} {st}')
print('total', tot)

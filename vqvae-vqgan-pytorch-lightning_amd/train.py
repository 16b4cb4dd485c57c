"""Training entry point: the run-configuration rules of the reference's ``vqvae/train.py`` over ``MiniTrainer``.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
        vqvae-vqgan-pytorch-lightning_amd/train.py --params_file example_confs/standard_vqvae.yaml --seed 0 ...

One process per GPU (the reference lets Lightning spawn them; here the launcher is ``torch.distributed.run`` and
WORLD_SIZE plays the role of ``num_nodes * gpus``).  Reproduced from the reference, by line:
  * :55-56  the YAML schema of ``example_confs/*.yaml`` (``get_model_conf``);
  * :59-63  ``batch_size_per_device = cumulative_bs // (num_nodes * gpus)``, ``lr = base_lr * sqrt(cumulative_bs / 256)``;
  * :86-98  ``image_size / ae_conf / q_conf / l_conf`` pass-through and the derived ``t_conf``;
  * :101-103, :139-140  adversarial detection and the ``batch_size_per_device % 4`` guard (minibatch-stddev groups of 4);
  * :106-114  resume through ``VQVAE.load_from_checkpoint(..., strict=False, init_cb=False)``.
Out of scope: the ffcv / folder data modules and wandb.  Batches come from ``--dataset_path`` when it is a
``.pt`` / ``.npy`` tensor file of images [M,3,S,S] in [0,1], otherwise they are synthetic U(0,1) (the benchmark's input).
"""
from __future__ import annotations

import argparse
import importlib
import math
import os
import sys

import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.basename(os.path.dirname(os.path.abspath(__file__)))


def get_model_conf(filepath: str) -> dict:
    """vqvae/common_utils.py:32-37"""
    with open(filepath, 'r', encoding='utf-8') as stream:
        return yaml.safe_load(stream)


def derive_run_config(conf: dict, world_size: int, overrides: dict | None = None) -> dict:
    """Everything ``train.py:55-103,139-140`` derives from a config file and the device count, as one dict:
    image_size, ae_conf, q_conf, l_conf, t_conf, batch_size_per_device, cumulative_batch_size, learning_rate, max_epochs,
    use_adversarial.  ``overrides``: optional ``{'quantizer.num_embeddings': 8192, 'loss.adversarial_params.start_epoch': 0,
    'training.cumulative_bs': 512, ...}`` applied to the loaded YAML first (BASELINE.json quotes some configs with a
    different codebook size than the YAML carries)."""
    conf = _deep_copy(conf)
    for dotted, value in (overrides or {}).items():
        node = conf
        *path, leaf = dotted.split('.')
        for key in path:
            node = node[key]
        node[leaf] = value
    if world_size < 1:
        raise ValueError('world_size must be >= 1')
    tr = conf['training']
    cumulative_batch_size = int(tr['cumulative_bs'])
    batch_size_per_device = cumulative_batch_size // world_size                      # train.py:59-60
    learning_rate = float(tr['base_lr']) * math.sqrt(cumulative_batch_size / 256)    # train.py:62-63
    l_conf = conf['loss'] if 'loss' in conf.keys() else None                         # train.py:89
    t_conf = {'lr': learning_rate, 'betas': tr['betas'], 'eps': tr['eps'], 'weight_decay': tr['weight_decay'],
              'warmup_epochs': tr['warmup_epochs'] if 'warmup_epochs' in tr.keys() else None,
              'decay_epochs': tr['decay_epochs'] if 'decay_epochs' in tr.keys() else None}       # train.py:90-96
    use_adversarial = (l_conf is not None and 'adversarial_params' in l_conf.keys()
                       and l_conf['adversarial_params'] is not None)                  # train.py:99-101
    if use_adversarial and batch_size_per_device % 4 != 0:                            # train.py:139-140
        raise RuntimeError('batch size per device must be divisible by 4! (due to stylegan discriminator forward pass)')
    if batch_size_per_device < 1:
        raise RuntimeError(f'cumulative_bs {cumulative_batch_size} is smaller than the number of devices {world_size}')
    return dict(image_size=int(conf['image_size']), ae_conf=conf['autoencoder'], q_conf=conf['quantizer'], l_conf=l_conf,
                t_conf=t_conf, batch_size_per_device=batch_size_per_device, cumulative_batch_size=cumulative_batch_size,
                learning_rate=learning_rate, max_epochs=int(tr['max_epochs']), use_adversarial=use_adversarial)


def _deep_copy(x):
    if isinstance(x, dict):
        return {k: _deep_copy(v) for k, v in x.items()}
    if isinstance(x, list):
        return [_deep_copy(v) for v in x]
    return x


def parse_args(argv=None):
    p = argparse.ArgumentParser(description=__doc__.split('\n')[0])
    p.add_argument('--params_file', type=str, required=True, help='yaml file with model params (example_confs/*.yaml)')
    p.add_argument('--dataset_path', type=str, default=None, help='.pt / .npy tensor of images [M,3,S,S] in [0,1]; '
                                                                  'omitted: synthetic U(0,1) batches')
    p.add_argument('--save_path', type=str, default=None, help='directory for checkpoints')
    p.add_argument('--save_every_n_epochs', type=int, default=1)
    p.add_argument('--run_name', type=str, default='run')
    p.add_argument('--seed', type=int, required=True)
    p.add_argument('--loading_path', type=str, default=None, help='checkpoint to resume from')
    p.add_argument('--num_nodes', type=int, default=1)
    p.add_argument('--max_epochs', type=int, default=None, help='override training.max_epochs')
    p.add_argument('--batches_per_epoch', type=int, default=None, help='synthetic data: steps per epoch')
    p.add_argument('--dtype', choices=['bf16', 'f32', 'bf16x3'], default='bf16',
                   help="bf16: throughput mode; f32: exact fp32 products (the reference mode); bf16x3: fp32 storage, every conv product as three bf16 products (parity-grade)")
    p.add_argument('--no-graph', action='store_true', help='eager launches instead of hipGraph replay')
    p.add_argument('--deterministic', action='store_true', help='Trainer(deterministic=True) of the reference (train.py:130): bit-reproducible gradients')
    p.add_argument('--optimizer_param_set', choices=['auto', 'all', 'reference'], default='auto',
                   help="'reference': the tensors (and order) the reference hands AdamW (vqvae/model.py:384-410) -- what a "
                        "reference checkpoint's optimizer state is indexed by; 'auto': every tensor, unless --loading_path "
                        "holds an optimizer state of the reference's size")
    p.add_argument('--set', action='append', default=[], metavar='KEY=VALUE',
                   help='override a config entry, e.g. --set quantizer.num_embeddings=8192')
    return p.parse_args(argv)


def detect_param_set(model_mod, ckpt_path: str, ctor_kw: dict) -> str:
    """'reference' when the checkpoint's first optimizer state lists as many parameters as the reference's AdamW would
    hold for this architecture (vqvae/model.py:384-410 drops encoder tensors shadowed by decoder names), else 'all'"""
    ckpt = torch.load(ckpt_path, map_location='cpu', weights_only=False)
    states = ckpt.get('optimizer_states') or []
    if not states:
        return 'all'
    saved = sum(len(g['params']) for g in states[0]['param_groups'])
    probe = model_mod.VQVAE(init_cb=False, load_loss=False, **dict(ctor_kw, optimizer_param_set='reference'))
    n_ref = sum(len(g) for g in probe.optimizer_groups())
    probe.optimizer_param_set = 'all'
    n_all = sum(len(g) for g in probe.optimizer_groups())
    if saved == n_ref and saved != n_all:
        return 'reference'
    return 'all'


def parse_overrides(pairs) -> dict:
    return {k: yaml.safe_load(v) for k, v in (item.split('=', 1) for item in pairs)}


def _batches(args, run, device, rank, world):
    b, s = run['batch_size_per_device'], run['image_size']
    if args.dataset_path:
        data = torch.load(args.dataset_path) if args.dataset_path.endswith('.pt') else \
            torch.from_numpy(__import__('numpy').load(args.dataset_path))
        data = data.float()
        per_epoch = data.shape[0] // (b * world)
        return [data[(i * world + rank) * b:(i * world + rank + 1) * b].to(device) for i in range(per_epoch)]
    g = torch.Generator().manual_seed(args.seed + 1000 * rank)
    n = args.batches_per_epoch or 8
    return [torch.rand(b, 3, s, s, generator=g).to(device) for _ in range(n)]


def main(argv=None):
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    args = parse_args(argv)
    trainer_mod = importlib.import_module(PKG + '.trainer')
    model_mod = importlib.import_module(PKG + '.model')
    rank, local, world = trainer_mod.init_distributed()
    if not torch.cuda.is_available():
        raise SystemExit('train.py needs an MI355X: the train step is HIP kernels only (no CPU fallback)')
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)
    run = derive_run_config(get_model_conf(args.params_file), world, parse_overrides(args.set))
    torch.manual_seed(args.seed)                                     # pl.seed_everything: identical replicas
    dtype = torch.bfloat16 if args.dtype == 'bf16' else 'bf16x3' if args.dtype == 'bf16x3' else torch.float32
    kw = dict(image_size=run['image_size'], ae_conf=run['ae_conf'], q_conf=run['q_conf'], l_conf=run['l_conf'],
              t_conf=run['t_conf'], compute_dtype=dtype)
    param_set = args.optimizer_param_set
    if param_set == 'auto':
        param_set = 'all'
        if args.loading_path is not None:
            param_set = detect_param_set(model_mod, args.loading_path, kw)
    kw['optimizer_param_set'] = param_set
    if args.loading_path is not None:                                # train.py:106-111
        model = model_mod.VQVAE.load_from_checkpoint(args.loading_path, strict=False, init_cb=False, load_loss=True, **kw)
    else:
        model = model_mod.VQVAE(init_cb=True, load_loss=True, **kw)
    model = model.to(device).train()
    if run['use_adversarial']:
        model.criterion.discriminator.compute_dtype = model.compute_dtype
        model.criterion.perceptual_loss.net.compute_dtype = model.compute_dtype
    batches = _batches(args, run, device, rank, world)
    if not batches:
        raise SystemExit(f'train.py: the dataset holds fewer than one batch per rank '
                         f'({run["batch_size_per_device"]} images x {world} ranks)')
    max_epochs = args.max_epochs or run['max_epochs']
    trainer = trainer_mod.MiniTrainer(max_epochs=max_epochs, num_training_batches=len(batches),
                                      deterministic=True if args.deterministic else None)
    trainer.attach(model)
    start_epoch = 0
    if args.loading_path is not None:
        ckpt = trainer.load_checkpoint(model, args.loading_path, strict=False)
        start_epoch = int(ckpt.get('epoch', -1)) + 1
    if rank == 0:
        print(f'[INFO] batch size per device: {run["batch_size_per_device"]}')
        print(f'[INFO] cumulative batch size (all devices): {run["cumulative_batch_size"]}')
        print(f'[INFO] final learning rate: {run["learning_rate"]}')
    model.on_train_start()
    graphed = False
    if not args.no_graph:
        try:
            trainer.capture(model, batches[0], warmup=1, preserve_state=True)   # the settling step must not train
            graphed = True
        except RuntimeError as exc:
            if rank == 0:
                print(f'[INFO] eager launches ({exc})')
    step = trainer.train_batch_graphed if graphed else trainer.train_batch
    loss = None
    for epoch in range(start_epoch, max_epochs):
        model.current_epoch = epoch
        for i, batch in enumerate(batches):
            loss = step(model, batch, i)
        model.on_train_epoch_end()
        importlib.import_module(PKG + '.ops').check_kernel_health()       # a kernel that gave up on a rendezvous = untrustworthy gradients: stop
        if rank == 0:
            print(f'[epoch {epoch}] loss {float(loss):.6f}', flush=True)
        if args.save_path and (epoch + 1) % args.save_every_n_epochs == 0:
            os.makedirs(os.path.join(args.save_path, args.run_name), exist_ok=True)
            trainer.save_checkpoint(model, os.path.join(args.save_path, args.run_name, f'epoch={epoch:02d}.ckpt'))
    model.on_train_end()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
    return float(loss) if loss is not None else None


if __name__ == '__main__':
    main()

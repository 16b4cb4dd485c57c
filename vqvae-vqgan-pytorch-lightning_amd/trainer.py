"""``MiniTrainer``: the slice of ``pytorch_lightning.Trainer`` the reference's train loop relies on
(vqvae/train.py:128-142 + the hook order of SURVEY 3.2), one process per GPU, data parallel by batch.

Where Lightning wraps the model in DDP (bucketed NCCL all-reduce during backward + a buffer broadcast),
this trainer issues ONE RCCL all-reduce of the flat gradient arena per optimizer step
(:meth:`FlatAdamW.all_reduce_grads`); the EMA quantizer all-reduces its own statistics."""
from __future__ import annotations

import os
from typing import Iterable

import torch
import torch.distributed as dist

from . import _native, ops


GAN_OPT_OVERLAP = _native.switch('VQK_GAN_OPT_OVERLAP', '1') != '0'     # VQ-GAN replay: the AE optimizer step beside the discriminator half


def init_distributed(backend: str | None = None) -> tuple[int, int, int]:
    """(rank, local_rank, world) from the torchrun environment; no-op for a single process."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    force = os.environ.get('VQK_FORCE_DIST') == '1'          # exercise the RCCL path on a single GPU (tests)
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'nccl':
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    if world > 1 and torch.cuda.is_available() and 'VQK_TILE_QUEUE' not in os.environ:
        # data parallel: a collective's kernel may hold CUs next to the persistent conv grids -- those draw their tiles from a
        # queue then (first tile static: no start-up latency; +0.1 ms per step without contention, -24 % kernel time with 16 CUs
        # held: profiles/round5_comm_probe.txt).  bench.py times the forms itself and overrides this per form.
        from . import _native
        _native.check(_native.lib().vqk_set_tuning(b'TILE_QUEUE', 1), 'set_tuning(TILE_QUEUE)')
    return rank, local, world


def _quiesce_collectives() -> None:
    """before a hipGraph capture: every collective issued so far has completed on the device AND the process group's watchdog
    thread (polls every 100 ms) has retired it -- the watchdog must not query RCCL events while this thread captures (HIP
    rejects a query of an event whose stream is capturing: the watchdog would terminate the process)"""
    from .optim import quiesce_collectives
    quiesce_collectives()                                    # device idle + every tracked Work handle reports completion
    if dist.is_available() and dist.is_initialized() and dist.get_backend() == 'nccl':
        import time
        time.sleep(0.3)                                      # (three watchdog polls: belt and braces for collectives issued around the helpers)


class MiniTrainer:
    def __init__(self, max_epochs: int = 1, num_training_batches: int | None = None, deterministic: bool | None = None):
        """``deterministic``: ``pl.Trainer(deterministic=True)`` of vqvae/train.py:130 -- ordered partial sums instead of atomics
        in every gradient accumulation of the step (ops.set_deterministic); None leaves the process-wide setting alone."""
        if deterministic is not None:
            ops.set_deterministic(deterministic)
        self.max_epochs = max_epochs
        self.num_training_batches = num_training_batches
        self.optimizers = []
        self.global_step = 0
        self.overlap_allreduce = self.OVERLAP_ALLREDUCE          # per trainer: bench.py times both forms on the same model

    def attach(self, model):
        model.trainer = self
        if self.num_training_batches is None:
            self.num_training_batches = 1
        opt = model.configure_optimizers()
        self.optimizers = list(opt[0]) if isinstance(opt, tuple) else [opt]
        return self.optimizers

    def train_batch(self, model, batch, batch_index: int):
        """hook order of one Lightning iteration with automatic optimisation"""
        opt = self.optimizers[0]
        model.on_train_batch_start(batch, batch_index)
        if not getattr(model, 'automatic_optimization', True):       # VQ-GAN: the module steps both optimizers itself
            loss = model.training_step(batch, batch_index)
            self.global_step += 1
            return loss
        opt.zero_grad()
        if self._use_split(model, opt):
            loss = self._split_step(model, opt, batch, batch_index)
        else:
            loss = model.training_step(batch, batch_index)
            loss.backward()
            ops.join_side_streams()
            self._finish_deferred(model, opt)
            opt.all_reduce_grads()
        opt.step()
        self.global_step += 1
        return loss

    # ------------------------------------------------------------------ gradient all-reduce under the backward
    # Lightning's DDP overlaps its bucketed all-reduce with the backward (vqvae/train.py:128).  Here the backward is cut at
    # the decoder's input (VQVAE.split_backward): the decoder's gradients -- laid out first in the arena -- are all-reduced
    # while the quantizer + encoder backward runs; the second all-reduce covers the rest.  Two collectives instead of one.
    OVERLAP_ALLREDUCE = os.environ.get('VQK_OVERLAP_ALLREDUCE', '1') != '0'

    def _use_split(self, model, opt) -> bool:
        return (self.overlap_allreduce and opt.collective_on() and getattr(opt, 'front_numel', 0) > 0
                and hasattr(model, 'split_backward') and getattr(model, 'automatic_optimization', True))

    @staticmethod
    def _backward_halves(model):
        """callables running the backward in stages: the decoder; the quantizer + the deep part of the encoder; and -- when the
        encoder was cut (VQVAE.split_encoder) -- its high-resolution head"""
        l2_loss, q_loss = model._backward_terms
        zq, dec_in = model._backward_cut
        enc_cut = getattr(model, '_encoder_cut', None)

        def first():
            l2_loss.backward()

        def second():
            tensors, grads = [zq], [dec_in.grad]
            if q_loss.requires_grad:
                tensors.append(q_loss)
                grads.append(torch.ones_like(q_loss))
            torch.autograd.backward(tensors, grads)
            dec_in.grad = None

        if enc_cut is None:
            return first, second
        x_pre, x_leaf = enc_cut

        def third():
            torch.autograd.backward([x_pre], [x_leaf.grad])
            x_leaf.grad = None
        return first, second, third

    @staticmethod
    def _ranges(opt, n_stages: int):
        """arena ranges whose gradients are complete after each stage of the backward"""
        total = opt.flat_g.numel()
        if n_stages == 3:
            return [(0, opt.front_numel), (opt.front_numel, opt.back_start), (opt.back_start, total)]
        return [(0, opt.front_numel), (opt.front_numel, total)]

    def _split_step(self, model, opt, batch, batch_index):
        model.split_backward = True
        try:
            loss = model.training_step(batch, batch_index)
        finally:
            model.split_backward = False
        stages = self._backward_halves(model)
        works = []
        for k, (stage, (lo, hi)) in enumerate(zip(stages, self._ranges(opt, len(stages)))):
            stage()
            ops.join_side_streams()
            if k == len(stages) - 1:
                self._finish_deferred(model, opt)
            if hi > lo:
                works.append(opt.all_reduce_range(lo, hi))         # async: runs under the next stage
        for w in works:
            if w is not None:
                w.wait()
        return loss

    @staticmethod
    def _finish_deferred(model, opt):
        q = getattr(model, 'quantizer', None)
        if getattr(q, 'defer_update', False):                       # EMA statistics left pending by the forward
            q.finish_update(force_collective=opt.force_collective)

    # ------------------------------------------------------------------ hipGraph replay of forward + backward
    def capture(self, model, example_batch, warmup: int = 3, preserve_state: bool = False):
        """Capture zero_grad + training_step + backward of ONE step into a hipGraph (shapes are static in
        training) and replay it afterwards: the step is ~600 short kernel launches, which is host-bound when
        issued one by one.  The gradient all-reduce and the AdamW launch stay outside the graph (one call
        each), so schedules (lr) remain ordinary host scalars.  ``warmup`` eager steps run first (they DO
        update the model) so that every lazy allocation / kernel attribute is settled before capture;
        ``preserve_state``: weights, buffers, optimizer moments and step counts are put back afterwards, so a graphed run
        takes exactly the optimizer steps an eager run takes (train.py)."""
        opt = self.optimizers[0]
        snap = self._snapshot(model) if preserve_state else None
        q = getattr(model, 'quantizer', None)
        if hasattr(q, 'enable_device_schedule'):
            # the Gumbel temperature / KL weight are scheduled per step (model.py:218-225): the kernels read them from a device
            # buffer that set_consts() refreshes, so a replay follows the schedule
            q.enable_device_schedule(example_batch.device)
        if not getattr(model, 'automatic_optimization', True):
            return self._capture_gan(model, example_batch, warmup, snap)
        self._static_in = example_batch.clone()
        # the settling steps run on the stream the capture will use: every per-(device, stream) workspace of ops.py (split-K
        # scratch, GroupNorm sums, deterministic-mode slices) exists before the capture starts -- a first use INSIDE the
        # capture would put the allocation in the graph's pool and its zero fill into every replay
        side = self._cap_stream = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            ops._stream()
            for i in range(warmup):
                self._eager_step(model, self._static_in, i)
        torch.cuda.current_stream().wait_stream(side)
        _quiesce_collectives()
        if snap is not None:
            self._restore(model, snap)
        # EMA quantizer: its statistics all-reduce must not sit inside the captured graph -- the update is deferred and
        # finished (collective + update kernel) after each replay
        self._deferred_q = getattr(model, 'quantizer', None) if hasattr(getattr(model, 'quantizer', None), 'defer_update') else None
        if self._deferred_q is not None:
            self._deferred_q.defer_update = True
        self._graph = torch.cuda.CUDAGraph()
        self._graph2 = self._graph3 = None
        # thread_local: the autograd worker thread and (multi-GPU) the RCCL watchdog thread issue runtime calls
        # of their own while this thread captures
        model.defer_usage_accumulation = True      # host-side bookkeeping stays out of the captured region
        split = self._use_split(model, opt)
        try:
            if split:
                # two graphs sharing one memory pool: [zero_grad, forward, decoder backward] and [quantizer + encoder backward];
                # the all-reduce of the decoder's arena range is issued between the two replays
                model.split_backward = True
                with torch.cuda.graph(self._graph, stream=side, capture_error_mode='thread_local'):
                    opt.zero_grad()
                    self._static_loss = model.training_step(self._static_in, 0)
                    stages = self._backward_halves(model)
                    stages[0]()
                    ops.join_side_streams()
                self._graph2 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self._graph2, pool=self._graph.pool(), stream=side, capture_error_mode='thread_local'):
                    stages[1]()
                    ops.join_side_streams()
                if len(stages) == 3:                      # the encoder's high-resolution head: the deep levels' range is reduced under it
                    self._graph3 = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(self._graph3, pool=self._graph.pool(), stream=side, capture_error_mode='thread_local'):
                        stages[2]()
                        ops.join_side_streams()
            else:
                with torch.cuda.graph(self._graph, stream=side, capture_error_mode='thread_local'):
                    opt.zero_grad()
                    self._static_loss = model.training_step(self._static_in, 0)
                    self._static_loss.backward()
                    ops.join_side_streams()
        finally:
            model.defer_usage_accumulation = False
            model.split_backward = False
        self._static_hist = model.quantizer.last_hist          # rewritten by every replay
        return self._graph

    # ------------------------------------------------------------------ VQ-GAN step (manual optimisation) as three graphs
    def _capture_gan(self, model, example_batch, warmup: int, snap=None):
        """model.py:244-264 under hipGraph replay: [AE half: zero_grad, forward, LPIPS + generator loss, backward] /
        [discriminator half] / [discriminator half with the R1 term] -- the two optimizer steps, the gradient all-reduces and
        the choice of the R1 variant (every ``r1_reg_every`` steps) stay on the host between the replays."""
        ae_opt, disc_opt = self.optimizers
        self._static_in = example_batch.clone()
        # forward_autoencoder / forward_discriminator branch ON THE HOST on `current_epoch >= adversarial_start_epoch`
        # (loss.py:121,143): the captured graphs hold ONE side of that branch.  The phase they were captured in is kept, and
        # _train_batch_gan_graphed re-captures when the epoch crosses the threshold (gumbel_vqgan.yaml: start_epoch 100).
        self._gan_phase = self._gan_adversarial(model)
        side = self._cap_stream = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            ops._stream()
            # both discriminator variants run once in the settling steps -- the R1 variant is the step with
            # (epoch * batches + index) % r1_every == 0: the indices are chosen for it at ANY epoch (a re-capture at the start of
            # the adversarial phase included), so no lazy pack / workspace allocation happens inside a capture
            crit0 = model.criterion
            every0 = crit0.r1_regularization_every if crit0.r1_regularization_cost is not None else 0
            first = (-(model.current_epoch * self.num_training_batches)) % every0 if every0 else 0
            for i in range(first, first + max(warmup, 2)):
                model.on_train_batch_start(self._static_in, i)
                model.training_step(self._static_in, i)
        torch.cuda.current_stream().wait_stream(side)
        _quiesce_collectives()
        if snap is not None:
            self._restore(model, snap)
        crit = model.criterion
        every = crit.r1_regularization_every if crit.r1_regularization_cost is not None else 0
        model.defer_usage_accumulation = True
        self._gan = {}
        try:
            g_ae = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g_ae, stream=side, capture_error_mode='thread_local'):
                res = model._gan_ae_half(self._static_in)
            self._gan['ae'] = (g_ae, res, model._gan_state[2])
            for key, step in (('d', 1), ('d_r1', 0)):
                if key == 'd_r1' and not every:
                    continue
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=g_ae.pool(), stream=side, capture_error_mode='thread_local'):
                    out = model._gan_disc_half(step if every else 1, retain_graph=True)
                self._gan[key] = (g, out)
        finally:
            model.defer_usage_accumulation = False
        self._gan_every = every
        self._static_hist = model.quantizer.last_hist
        self._graph = g_ae
        self._graph2 = self._graph3 = None
        return g_ae

    @staticmethod
    def _gan_adversarial(model) -> bool:
        return model.current_epoch >= model.criterion.adversarial_start_epoch

    def _train_batch_gan_graphed(self, model, batch, batch_index: int):
        ae_opt, disc_opt = self.optimizers
        if self._gan_adversarial(model) != self._gan_phase:
            # the adversarial phase starts (or, on a resumed / rewound run, ends): the graphs of the other phase would keep
            # replaying a step without generator loss and never step the discriminator.  Capture this phase's graphs; the
            # settling steps of the capture do not train (state snapshot put back).
            self._gan = None
            self._capture_gan(model, batch, warmup=2, snap=self._snapshot(model))
        model.on_train_batch_start(batch, batch_index)
        if batch is not self._static_in:
            self._static_in.copy_(batch, non_blocking=True)
        ops.repack_owned(None)
        g_ae, res, q_loss = self._gan['ae']
        g_ae.replay()
        step = model.current_epoch * self.num_training_batches + batch_index
        key = 'd_r1' if (self._gan_every and step % self._gan_every == 0) else 'd'
        g, (loss, d_loss, r1_penalty) = self._gan[key]
        if GAN_OPT_OVERLAP and loss is not None:
            # the autoencoder's all-reduce + AdamW + operand refresh beside the discriminator half: that graph reads the
            # reconstruction the AE half left, the real batch and the discriminator -- nothing the AE optimizer writes (its
            # backward stops at the discriminator's parameters, model.py::_gan_disc_half); model.py:251-264 has the same order of
            # effects (opt_ae.step() before the discriminator loss, which sees x_rec.detach() of the OLD weights)
            cur, osd = torch.cuda.current_stream(), ops.aux_stream(self._static_in.device, 'ae_opt')
            osd.wait_stream(cur)
            with torch.cuda.stream(osd):
                ae_opt.all_reduce_grads()
                ae_opt.step()
            g.replay()
            cur.wait_stream(osd)
        else:
            ae_opt.all_reduce_grads()
            ae_opt.step()
            g.replay()
        if loss is not None:
            disc_opt.all_reduce_grads()
            disc_opt.step()
        model._gan_log(res, q_loss, d_loss, r1_penalty)
        model.accumulate_usage(self._static_hist)
        self.global_step += 1
        return res[0]

    def _snapshot(self, model):
        return dict(state={k: v.detach().clone() for k, v in model.state_dict().items()},
                    opts=[(o.flat_v.clone(), None if o.flat_m is None else o.flat_m.clone(), o.step_count) for o in self.optimizers],
                    usage=(None if getattr(model, 'train_epoch_usage_count', None) is None else model.train_epoch_usage_count.clone()),
                    step=self.global_step,
                    # the settling steps draw Gumbel noise / augmentation boxes and log: neither may leak into the run
                    rng=(torch.cuda.get_rng_state() if torch.cuda.is_available() else None, torch.get_rng_state()),
                    logged=(dict(model.logged) if isinstance(getattr(model, 'logged', None), dict) else None))

    @torch.no_grad()
    def _restore(self, model, snap):
        own = model.state_dict()
        for k, v in snap['state'].items():
            own[k].copy_(v)                                          # in place: parameters stay views of the flat arena
        for o, (v, m, n) in zip(self.optimizers, snap['opts']):
            o.flat_v.copy_(v)
            if m is not None:
                o.flat_m.copy_(m)
            o.step_count = n
            o.generation += 1
            ops.repack_owned(o)
            if o.shadow is not None:
                o.shadow.copy_(o.flat_p)
        if hasattr(model, 'train_epoch_usage_count'):
            model.train_epoch_usage_count = snap['usage']
        self.global_step = snap['step']
        if snap.get('rng') is not None:
            if snap['rng'][0] is not None:
                torch.cuda.set_rng_state(snap['rng'][0])
            torch.set_rng_state(snap['rng'][1])
        if snap.get('logged') is not None:
            model.logged = snap['logged']

    def _eager_step(self, model, batch, batch_index):
        opt = self.optimizers[0]
        opt.zero_grad()
        if self._use_split(model, opt):
            loss = self._split_step(model, opt, batch, batch_index)
        else:
            loss = model.training_step(batch, batch_index)
            loss.backward()
            ops.join_side_streams()
            self._finish_deferred(model, opt)
            opt.all_reduce_grads()
        opt.step()
        return loss

    def train_batch_graphed(self, model, batch, batch_index: int):
        if not getattr(model, 'automatic_optimization', True):
            return self._train_batch_gan_graphed(model, batch, batch_index)
        opt = self.optimizers[0]
        model.on_train_batch_start(batch, batch_index)
        if batch is not self._static_in:
            self._static_in.copy_(batch, non_blocking=True)
        ops.repack_owned(None)               # operands of weights changed outside the optimizer (normally none)
        self._graph.replay()
        if self._graph2 is not None:
            g3 = getattr(self, '_graph3', None)
            ranges = self._ranges(opt, 3 if g3 is not None else 2)
            works = [opt.all_reduce_range(*ranges[0])]     # the decoder's gradients, under the encoder's backward
            self._graph2.replay()
            if g3 is not None:
                works.append(opt.all_reduce_range(*ranges[1]))     # quantizer + deep encoder levels, under the encoder head's backward
                g3.replay()
            model.accumulate_usage(self._static_hist)
            self._finish_deferred(model, opt)
            if ranges[-1][1] > ranges[-1][0]:
                works.append(opt.all_reduce_range(*ranges[-1]))
            for w in works:
                if w is not None:
                    w.wait()
        else:
            model.accumulate_usage(self._static_hist)      # epoch code histogram: eager add of the replay's histogram
            self._finish_deferred(model, opt)
            opt.all_reduce_grads()
        opt.step()
        self.global_step += 1
        return self._static_loss

    # ------------------------------------------------------------------ checkpoints (vqvae/train.py:106-122)
    def save_checkpoint(self, model, path: str) -> None:
        """the keys of a Lightning checkpoint that the reference's ``load_from_checkpoint`` / ``state_dict`` consumers
        read: 'state_dict' (plain contiguous tensors under the LightningModule's own names), 'optimizer_states' (loadable
        by ``torch.optim.AdamW``), 'epoch', 'global_step'.  NOT a full Lightning resume file ('loops', 'callbacks',
        'lr_schedulers' and the version stamp are absent: ``Trainer.fit(ckpt_path=...)`` of real Lightning needs those).
        Rank 0 writes (temporary file + rename), every rank waits."""
        import torch.distributed as dist
        ops.check_kernel_health()                       # never persist weights stepped on a corrupted gradient
        distributed = dist.is_available() and dist.is_initialized()
        if not distributed or dist.get_rank() == 0:
            sd = {k: v.detach().clone(memory_format=torch.contiguous_format).cpu() for k, v in model.state_dict().items()}
            adamw_defaults = dict(amsgrad=False, maximize=False, foreach=None, capturable=False, differentiable=False,
                                  fused=None, decoupled_weight_decay=True)
            opts = [{'state': {i: {k: (t.cpu() if torch.is_tensor(t) else t) for k, t in e.items()}
                               for i, e in o.state_dict()['state'].items()},
                     'param_groups': [{**adamw_defaults, **g} for g in o.state_dict()['param_groups']]}
                    for o in self.optimizers]
            tmp = f'{path}.tmp.{os.getpid()}'
            torch.save({'epoch': int(getattr(model, 'current_epoch', 0)), 'global_step': int(self.global_step),
                        'state_dict': sd, 'optimizer_states': opts}, tmp)
            os.replace(tmp, path)
        if distributed:
            from .optim import barrier
            barrier()

    def load_checkpoint(self, model, path: str, strict: bool = True) -> dict:
        """resume: weights, optimizer moments / step counts, epoch and global step (call after ``attach``)"""
        ckpt = torch.load(path, map_location='cpu', weights_only=False)
        with torch.no_grad():
            own = model.state_dict()
            missing = [k for k in own if k not in ckpt['state_dict']]
            if strict and missing:
                raise KeyError(f'checkpoint lacks {missing[:4]}...')
            for k, v in ckpt['state_dict'].items():
                if k in own:
                    own[k].copy_(v.to(own[k].device))               # in place: parameters stay views of the flat arena
        for o, s in zip(self.optimizers, ckpt.get('optimizer_states', [])):
            o.load_state_dict(s)
            o.generation += 1
            ops.repack_owned(o)                                      # cached conv operands follow the loaded weights
            if o.shadow is not None:
                o.shadow.copy_(o.flat_p)
        model.current_epoch = ckpt.get('epoch', 0)
        self.global_step = ckpt.get('global_step', 0)
        return ckpt

    def fit(self, model, batches: Iterable):
        batches = list(batches)
        if self.num_training_batches is None:
            self.num_training_batches = len(batches)
        if not self.optimizers:
            self.attach(model)
        model.train()
        model.on_train_start()
        loss = None
        for epoch in range(self.max_epochs):
            model.current_epoch = epoch
            for i, batch in enumerate(batches):
                loss = self.train_batch(model, batch, i)
            model.on_train_epoch_end()
            ops.check_kernel_health()                   # (the epoch end synchronises for the code-usage statistics anyway)
        model.on_train_end()
        return loss

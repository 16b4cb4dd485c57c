"""``VQVAE``: the reference's LightningModule surface (vqvae/model.py:23-562) over the vqk kernels.

Kept verbatim from the reference: constructor signature and config dictionaries (:25-77), attribute
names (``encoder``, ``decoder``, ``quantizer``, ``criterion``), ``forward`` return convention (:151-161),
hook names (:163, :202, :232, :297, :305, :372) and the inference API (:458-489), so the class can be
handed to a Lightning ``Trainer`` when ``pytorch_lightning`` is installed, or to the bundled
:class:`~.trainer.MiniTrainer` when it is not.

Deliberate departures (reference defects):
  * ``training_step`` returns the auto-encoder loss (the reference returns an unbound local on the MSE path);
  * the epoch usage count is accumulated (the reference overwrites it with a unary plus);
  * ``configure_optimizers`` by default optimises all 144 tensors; ``optimizer_param_set='reference'``
    reproduces the reference's name-collision (encoder tensors shadowed by same-named decoder tensors);
  * ``on_train_end`` tolerates a missing scheduler;
  * no host synchronisation inside the step: scalars stay on the device.
"""
from __future__ import annotations

from typing import Any

import os

import torch
import torch.distributed as dist
from torch import nn

from . import ops
from .modules.abstract_modules.base_autoencoder import BaseVQVAE
from .modules.autoencoder import Decoder, Encoder, GroupNorm, Conv2d, resolve_compute_dtype, set_compute_dtype
from .modules.vector_quantizers import (EMAVectorQuantizer, EntropyVectorQuantizer, GumbelVectorQuantizer,
                                        VectorQuantizer)
from .modules.loss import loss as loss_mod
from .modules.loss.loss import VQLPIPSWithDiscriminator
from .optim import FlatAdamW
from .schedulers import CosineScheduler, LinearCosineScheduler, LinearScheduler

try:                                                    # optional: real Lightning when present
    import pytorch_lightning as pl
    _LightningBase = pl.LightningModule
except Exception:                                       # pragma: no cover - not installed in this image
    class _LightningBase(nn.Module):
        """duck-typed stand-in for ``pl.LightningModule``: the attributes the hooks touch"""

        def __init__(self):
            super().__init__()
            self.trainer = None
            self.current_epoch = 0
            self.automatic_optimization = True
            self.logged = {}

        def log(self, name, value, **_):
            self.logged[name] = value

        def optimizers(self):
            return self.trainer.optimizers if len(self.trainer.optimizers) > 1 else self.trainer.optimizers[0]

        def manual_backward(self, loss):
            loss.backward()

        @classmethod
        def load_from_checkpoint(cls, checkpoint_path, map_location=None, strict: bool = True, **kwargs):
            """``pl.LightningModule.load_from_checkpoint`` as vqvae/train.py:106-111 and evaluate.py:49 call it: a
            Lightning checkpoint is a ``torch.save``d dict with the weights under 'state_dict'"""
            ckpt = torch.load(checkpoint_path, map_location=map_location or 'cpu', weights_only=False)
            model = cls(**kwargs)
            model.load_state_dict(ckpt['state_dict'] if 'state_dict' in ckpt else ckpt, strict=strict)
            return model


class VQVAE(BaseVQVAE, _LightningBase):

    def __init__(self, image_size: int, ae_conf: dict, q_conf: dict, l_conf: dict | None, t_conf: dict | None,
                 init_cb: bool = True, load_loss: bool = True, compute_dtype: torch.dtype = torch.float32,
                 optimizer_param_set: str = 'all', training_augmentation: bool = False):
        super().__init__(image_size=image_size)
        self.t_conf = t_conf
        # True: the reference's training behaviour (random resized crop + flip before every training step,
        # base_autoencoder.py:44-48); False (default): the identity the golden vectors and the benchmark are defined on
        self.training_augmentation = training_augmentation
        self.cb_size = q_conf['num_embeddings']
        self.latent_dim = q_conf['embedding_dim']
        self.reinit_every_n_epochs = q_conf['reinit_every_n_epochs']
        self.optimizer_param_set = optimizer_param_set
        self.defer_usage_accumulation = False
        self.split_backward = False            # MiniTrainer (data parallel): backward cut at the decoder's input ...
        self.split_encoder = os.environ.get('VQK_SPLIT_ENCODER', '1') != '0'     # ... and behind the encoder's high-resolution head
        self._backward_cut = self._backward_terms = self._encoder_cut = None
        self.kl_warmup_epochs = self.temp_decay_epochs = self.temp_final = None

        qt, qp = q_conf['type'], q_conf['params']
        if qt == 'standard':
            self.quantizer = VectorQuantizer(self.cb_size, self.latent_dim, float(qp['commitment_cost']))
        elif qt == 'ema':
            self.quantizer = EMAVectorQuantizer(self.cb_size, self.latent_dim, float(qp['commitment_cost']),
                                                float(qp['decay']), float(qp['epsilon']))
        elif qt == 'entropy':
            self.quantizer = EntropyVectorQuantizer(self.cb_size, self.latent_dim, float(qp['ent_loss_ratio']),
                                                    float(qp['ent_temperature']), str(qp['ent_loss_type']),
                                                    float(qp['commitment_cost']))
        elif qt == 'gumbel':
            self.quantizer = GumbelVectorQuantizer(self.cb_size, self.latent_dim, bool(qp['straight_through']),
                                                   float(qp['temp']), float(qp['kl_cost']))
            self.kl_warmup_epochs = qp.get('kl_warmup_epochs')
            self.temp_decay_epochs = qp.get('temp_decay_epochs')
            self.temp_final = qp.get('temp_final')
        else:
            raise ValueError(f'unrecognized quantizer: {qt}')

        ch, nrb, mult = ae_conf['channels'], ae_conf['num_res_blocks'], tuple(ae_conf['channel_multipliers'])
        self.encoder = Encoder(ch, nrb, mult, self.cb_size if qt == 'gumbel' else self.latent_dim)   # model.py:130
        self.decoder = Decoder(ch, nrb, mult, self.latent_dim)

        if load_loss:
            if l_conf is None:
                self.criterion = MSELoss()
            elif l_conf['adversarial_params'] is None:
                raise NotImplementedError('VQLPIPS (AlexNet ablation, loss.py:167-199) is out of scope (SURVEY 2 row 7)')
            else:
                self.criterion = VQLPIPSWithDiscriminator(image_size, l_conf['l1_weight'], l_conf['l2_weight'],
                                                          l_conf['perc_weight'], l_conf['adversarial_params'])
        else:
            self.criterion = None

        if init_cb:
            self.quantizer.init_codebook()
        self.compute_dtype, self.conv_products = resolve_compute_dtype(compute_dtype)
        set_compute_dtype(self, compute_dtype)
        if self.conv_products != 'fp32' and self.criterion is not None:
            # the split-product mode is built and pinned for the AUTOENCODER's convs (tests/test_gpu_fullsize.py); the loss networks
            # (LPIPS-VGG16, the StyleGAN2 discriminator) keep exact fp32 products in the fp32 storage modes
            for m in self.criterion.modules():
                if hasattr(m, 'conv_products'):
                    m.conv_products = 'fp32'

    # ------------------------------------------------------------------ forward (model.py:151-161)
    def forward(self, x: torch.Tensor):
        z = self.encoder(x)
        quantized, used_indices, e_loss = self.quantizer(z)
        x_recon = self.decoder(quantized)
        return x_recon, e_loss, used_indices

    # ------------------------------------------------------------------ schedules (model.py:163-230)
    def on_train_start(self):
        lr = float(self.t_conf['lr'])
        nb = self.trainer.num_training_batches
        wu, de = self.t_conf.get('warmup_epochs'), self.t_conf.get('decay_epochs')
        if wu is not None and de is not None:
            self.scheduler = LinearCosineScheduler(0, de * nb, lr, lr / 2., wu * nb)
        elif wu is not None:
            self.scheduler = LinearScheduler(0, wu * nb, 1e-20, lr)
        elif de is not None:
            self.scheduler = CosineScheduler(0, de * nb, lr, lr / 2.)
        if isinstance(self.quantizer, GumbelVectorQuantizer):                  # model.py:189-200
            temp, kl = self.quantizer.get_consts()
            if self.kl_warmup_epochs is not None:
                self.quantizer.kl_warmup = CosineScheduler(0, int(self.kl_warmup_epochs * nb), 0.0, kl)
            if self.temp_decay_epochs is not None and self.temp_final is not None:
                self.quantizer.temp_decay = CosineScheduler(0, int(self.temp_decay_epochs * nb), temp, self.temp_final)

    def on_train_batch_start(self, _: Any, batch_index: int):
        step = self.current_epoch * self.trainer.num_training_batches + batch_index
        step_lr = self.scheduler.step(step) if self.scheduler is not None else self.t_conf['lr']
        for optimizer in self.trainer.optimizers:
            for g in optimizer.param_groups:
                g['lr'] = step_lr
        this_temp, this_kl = 0.0, 0.0
        if isinstance(self.quantizer, GumbelVectorQuantizer):                  # model.py:218-225
            this_temp, this_kl = self.quantizer.get_consts()
            if self.quantizer.kl_warmup is not None:
                this_kl = self.quantizer.kl_warmup.step(step)
            if self.quantizer.temp_decay is not None:
                this_temp = self.quantizer.temp_decay.step(step)
            self.quantizer.set_consts(this_temp, this_kl)
        self.log('gumbel_quantizer/temperature', this_temp, sync_dist=True)
        self.log('gumbel_quantizer/kl_constant', this_kl, sync_dist=True)

    # ------------------------------------------------------------------ the step (model.py:232-295)
    def _preprocess_train(self, images, training: bool):
        """fused clamp / normalise / NHWC pad; with ``self.training_augmentation`` also the reference's RandomResizedCrop +
        RandomHorizontalFlip (base_autoencoder.py:20-22,44-48) in the same kernel, drawn on the device"""
        if training and self.training_augmentation:
            n, _, h, w = images.shape
            box, flip = ops.random_crop_params(n, h, w, images.device)
            return ops.raw_augment_preprocess(images, box, flip, self.compute_dtype, want_target=True)
        return ops.raw_preprocess(images, self.compute_dtype, want_target=True)

    def _step_losses(self, batch, training: bool):
        images = batch[0] if isinstance(batch, (tuple, list)) else batch
        x_pad, target = self._preprocess_train(images, training)                          # clamp, normalise, NHWC
        enc_split = self._encoder_split() if (training and self.split_backward and self.split_encoder) else None
        z = self.encoder(x_pad, cut_after=enc_split) if enc_split is not None else self.encoder(x_pad)
        self._encoder_cut = self.encoder.last_cut if enc_split is not None else None
        quantized, used_indices, q_loss = self.quantizer(z)
        dec_in = quantized
        if training and self.split_backward:
            # cut the autograd graph at the decoder's input: the trainer runs the decoder's backward first, starts the
            # all-reduce of the decoder's gradients and runs the quantizer + encoder backward underneath it
            dec_in = quantized.detach().requires_grad_(True)
            self._backward_cut = (quantized, dec_in)
        recon_pad = self.decoder.forward_padded(dec_in)
        l2_loss = ops.mse_loss(recon_pad, target, true_channels=3)
        return recon_pad, used_indices, q_loss, l2_loss

    def _gan_ae_half(self, batch: Any):
        """first half of model.py:244-264: zero the AE gradients, forward, nll + g_weight * g_loss + q_loss, backward"""
        images = batch[0] if isinstance(batch, (tuple, list)) else batch
        x_pad, target = self._preprocess_train(images, True)
        z = self.encoder(x_pad)
        quantized, _, q_loss = self.quantizer(z)
        recon_pad = self.decoder.forward_padded(quantized)
        ae_opt, _ = self.optimizers()
        ae_opt.zero_grad()
        # the generator loss needs the discriminator's INPUT gradient only: its weight / bias gradients from this half are thrown
        # away by the reference too (model.py:258 zeroes them before the discriminator step) -- not computing them saves every
        # weight-gradient kernel of one of the step's three discriminator passes
        disc_params = [p for p in self.criterion.discriminator.parameters() if p.requires_grad]
        share = loss_mod.SHARE_FAKE_PASS and self.current_epoch >= self.criterion.adversarial_start_epoch
        if share:
            # the discriminator half reuses THIS pass over the reconstruction (loss.SHARE_FAKE_PASS): its parameters stay in the
            # graph, the generator's backward skips their gradients (ops.no_param_grads) and keeps the saved activations
            self.criterion._share(None, None)
            res = self.criterion.forward_autoencoder(q_loss, target, recon_pad, self.current_epoch,
                                                     last_layer=self.decoder.conv_out.weight)
            with ops.no_param_grads():
                ops.backward(res[0], retain_graph=True)     # (the discriminator half backpropagates through the shared D(fake) pass)
        else:
            for p in disc_params:
                p.requires_grad_(False)
            try:
                res = self.criterion.forward_autoencoder(q_loss, target, recon_pad, self.current_epoch,
                                                         last_layer=self.decoder.conv_out.weight)
            finally:
                for p in disc_params:
                    p.requires_grad_(True)
            self.manual_backward(res[0])
        self.criterion.join_aux_streams()                   # LPIPS ran (and was differentiated) on its own stream
        self._gan_state = (target, recon_pad, q_loss, res)
        return res

    def _gan_disc_half(self, step: int, retain_graph: bool = False):
        """second half: discriminator loss (+ R1 every r1_reg_every steps) and its backward; (loss, d_loss, r1_penalty).
        ``retain_graph``: only the graph-capturing trainer needs the generator half's autograd graph afterwards (it captures this
        half twice on one generator half, with and without R1); an eager step releases it (MiniTrainer-less callers included)."""
        target, recon_pad, _, _ = self._gan_state
        _, disc_opt = self.optimizers()
        loss, d_loss, r1_penalty = self.criterion.forward_discriminator(target, recon_pad, self.current_epoch, step)
        if loss is not None:
            disc_opt.zero_grad()
            if getattr(self.criterion, 'shared_fake_logits', None) is not None:
                # the fake half of the loss hangs on the generator half's graph: only the discriminator's leaves are wanted
                loss.backward(inputs=[p for p in self.criterion.discriminator.parameters() if p.requires_grad], retain_graph=retain_graph)
            else:
                self.manual_backward(loss)
            self.criterion.join_aux_streams()               # the real pass (forward + backward) ran on its own stream
        if not retain_graph:
            # nothing of the step's autograd graph outlives the step: the saved activations of the generator half (encoder, decoder,
            # both VGG passes, the D(fake) pass) would otherwise stay allocated until the NEXT step's forward has allocated its own
            # set -- twice the activation memory at the peak
            self.criterion._share(None, None)
            t, _, q, res = self._gan_state
            self._gan_state = (None, None, q.detach(), tuple(v.detach() if torch.is_tensor(v) else v for v in res))
            loss = loss.detach() if loss is not None else None
        return loss, d_loss, r1_penalty

    def _gan_log(self, res, q_loss, d_loss, r1_penalty):
        ae_loss, l1_loss, l2_loss, p_loss, g_loss, g_weight = res
        for name, value in (('g_weight', g_weight), ('r1_penalty', r1_penalty)):          # model.py:277-278
            self.log(name, value.detach() if torch.is_tensor(value) else value, sync_dist=True, on_step=False, on_epoch=True)
        for name, value in (('train/loss', ae_loss), ('train/l1_loss', l1_loss), ('train/l2_loss', l2_loss),
                            ('train/quant_loss', q_loss), ('train/perc_loss', p_loss), ('train/gen_loss', g_loss),
                            ('train/disc_loss', d_loss)):
            self.log(name, value.detach(), sync_dist=True, on_step=False, on_epoch=True)

    def _gan_training_step(self, batch: Any, batch_index: int):
        """manual optimisation, model.py:244-264: AE step (nll + g_weight * g_loss + q_loss), then discriminator step"""
        res = self._gan_ae_half(batch)
        ae_opt, disc_opt = self.optimizers()
        ae_opt.all_reduce_grads()
        ae_opt.step()
        step = self.current_epoch * self.trainer.num_training_batches + batch_index
        loss, d_loss, r1_penalty = self._gan_disc_half(step)
        if loss is not None:
            disc_opt.all_reduce_grads()
            disc_opt.step()
        self._gan_log(res, self._gan_state[2], d_loss, r1_penalty)
        self.accumulate_usage(self.quantizer.last_hist)
        return res[0].detach()

    def training_step(self, batch: Any, batch_index: int):
        if isinstance(self.criterion, VQLPIPSWithDiscriminator):
            return self._gan_training_step(batch, batch_index)
        _, used_indices, q_loss, l2_loss = self._step_losses(batch, training=True)
        ae_loss = q_loss + l2_loss
        if self.split_backward:
            self._backward_terms = (l2_loss, q_loss)       # d(ae_loss)/d(l2_loss) = d(ae_loss)/d(q_loss) = 1
        for name, value in (('train/loss', ae_loss), ('train/l2_loss', l2_loss), ('train/quant_loss', q_loss)):
            self.log(name, value.detach(), sync_dist=True, on_step=False, on_epoch=True)      # device scalars: no sync
        self.accumulate_usage(self.quantizer.last_hist)
        return ae_loss

    def accumulate_usage(self, hist: torch.Tensor) -> None:
        """epoch code histogram += this step's (model.py:289-293; the reference's ``else + used_indices`` keeps only the
        last batch).  Inside a hipGraph capture this must not run: a captured ``a = a + hist`` would re-read the
        capture-time tensor on every replay, so the trainer sets ``defer_usage_accumulation`` while capturing and calls
        this itself after each replay with the graph's static histogram."""
        if self.defer_usage_accumulation:
            return
        self.train_epoch_usage_count = hist.clone() if self.train_epoch_usage_count is None \
            else self.train_epoch_usage_count + hist

    def on_train_epoch_end(self):
        if (self.reinit_every_n_epochs is not None and self.current_epoch % self.reinit_every_n_epochs == 0
                and self.current_epoch > 0 and self.train_epoch_usage_count is not None):
            count = self.train_epoch_usage_count.float()
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                dist.all_reduce(count, op=dist.ReduceOp.SUM)       # every replica re-initialises from the GLOBAL usage
            usage = self.quantizer.get_codebook_usage(count)[0]
            self.quantizer.reinit_unused_codes(usage)
        self.train_epoch_usage_count = None

    def on_train_end(self):
        if self.scheduler is not None:
            self.scheduler.destroy()

    @torch.no_grad()
    def validation_step(self, batch: Any, batch_index: int):
        _, _, q_loss, l2_loss = self._step_losses(batch, training=False)
        loss = q_loss + l2_loss
        self.log('validation/loss', loss, sync_dist=True, on_step=False, on_epoch=True)
        hist = self.quantizer.last_hist
        self.val_epoch_usage_count = hist.clone() if self.val_epoch_usage_count is None \
            else self.val_epoch_usage_count + hist
        return loss

    def on_validation_epoch_end(self):
        if self.val_epoch_usage_count is not None:
            _, perplexity, cb_usage = self.quantizer.get_codebook_usage(self.val_epoch_usage_count.float())
            self.log('val_metrics/used_codebook', cb_usage, sync_dist=True)
            self.log('val_metrics/perplexity', perplexity, sync_dist=True)
        self.val_epoch_usage_count = None

    # ------------------------------------------------------------------ test loop (model.py:491-553)
    def on_test_epoch_start(self):
        from .metrics import ReconstructionMetrics
        self.test_metrics = ReconstructionMetrics(next(self.parameters()).device)
        self.test_usage_count = None

    @torch.no_grad()
    def test_step(self, images: Any, _):
        """reconstructions in [0, 1] against the (clamped) inputs: MSE / PSNR / SSIM and the code histogram, all on the
        device.  rFID (model.py:535-541) needs the pretrained Inception network and is not computed offline."""
        images = images[0] if isinstance(images, (tuple, list)) else images
        recon, _, used_indices = self(self.preprocess_batch(images))
        recon = self.preprocess_visualization(recon.float())[:, :3]
        hist = torch.bincount(used_indices.reshape(-1), minlength=self.cb_size)
        # (the reference writes `else + used_indices`, i.e. keeps the LAST batch's histogram; the sum is what it logs as usage)
        self.test_usage_count = hist if self.test_usage_count is None else self.test_usage_count + hist
        self.test_metrics.update(recon, images.to(recon.device).float())

    def on_test_epoch_end(self):
        out = self.test_metrics.compute()
        for name in ('mse', 'ssim', 'psnr'):
            self.log(name, out[name], sync_dist=True)
        _, perplexity, cb_usage = self.quantizer.get_codebook_usage(self.test_usage_count.float())
        self.log('used_codebook', cb_usage, sync_dist=True)
        self.log('perplexity', perplexity, sync_dist=True)
        out.update(used_codebook=cb_usage, perplexity=perplexity)
        return out

    # ------------------------------------------------------------------ optimizer (model.py:372-440)
    def optimizer_groups(self):
        """(decay, no_decay) lists of (full name, parameter) IN OPTIMIZER ORDER.  decay = conv weights; no decay =
        biases, GroupNorm affine, codebook (model.py:388-396, :424-425).  The reference keys its sets by the name
        relative to encoder / decoder / quantizer and sorts by that key (model.py:384-410): with
        ``optimizer_param_set='reference'`` the later sub-module shadows an earlier one of the same relative name (91 of
        144 tensors survive) and the order is the sorted relative names -- what a reference 'optimizer_states' checkpoint
        is indexed by.  'all' keeps every tensor, sorted by full name."""
        seen = {}
        for prefix, sub in (('encoder', self.encoder), ('decoder', self.decoder), ('quantizer', self.quantizer)):
            for mn, m in sub.named_modules():
                for pn, p in m.named_parameters(recurse=False):
                    if not p.requires_grad and self.optimizer_param_set != 'reference':
                        continue              # (the reference hands the frozen EMA codebook to AdamW too: it never gets a grad)
                    rel = f'{mn}.{pn}' if mn else pn
                    is_decay = pn.endswith('weight') and isinstance(m, Conv2d)
                    key = rel if self.optimizer_param_set == 'reference' else f'{prefix}.{rel}'
                    seen[key] = (f'{prefix}.{rel}', p, is_decay)      # 'reference': later sub-modules shadow earlier ones
        decay = [(full, p) for key, (full, p, d) in sorted(seen.items()) if d]
        no_decay = [(full, p) for key, (full, p, d) in sorted(seen.items()) if not d]
        return decay, no_decay

    def _encoder_split(self):
        if not hasattr(self, '_enc_split_cache'):
            fn = getattr(self.encoder, 'shallow_split', None)
            frac = float(os.environ.get('VQK_SPLIT_ENCODER_FRACTION', '0.12'))
            self._enc_split_cache = fn(frac) if (fn is not None and self.split_encoder) else None
        return self._enc_split_cache

    def configure_optimizers(self):
        lr = float(self.t_conf['lr'])
        betas = [float(b) for b in self.t_conf['betas']]
        eps, wd = float(self.t_conf['eps']), float(self.t_conf['weight_decay'])
        decay, no_decay = self.optimizer_groups()
        groups = [{'params': [p for _, p in decay], 'weight_decay': wd},
                  {'params': [p for _, p in no_decay], 'weight_decay': 0.0}]
        # the decoder's tensors come first in the gradient arena: their all-reduce starts while the encoder's backward runs
        # ... and the encoder's high-resolution head (few parameters, ready last) at the very end: the only range whose
        # all-reduce is exposed (trainer.MiniTrainer._split_step)
        front = {id(p) for p in self.decoder.parameters()}
        split = self._encoder_split()
        back = {id(p) for p in self.encoder.shallow_parameters(split)} if split is not None else set()
        ae_optimizer = FlatAdamW(groups, lr=lr, betas=betas, eps=eps, weight_decay=wd, arena_front=front, arena_back=back)
        if isinstance(self.criterion, VQLPIPSWithDiscriminator):                      # model.py:431-438
            disc_optimizer = FlatAdamW(list(self.criterion.discriminator.parameters()), lr=lr, betas=betas, eps=eps,
                                       weight_decay=wd)
            self.automatic_optimization = False
            return [ae_optimizer, disc_optimizer], []
        return ae_optimizer

    # ------------------------------------------------------------------ inference API (model.py:458-489)
    @torch.no_grad()
    def get_tokens(self, images: torch.Tensor) -> torch.Tensor:
        return self.quantizer.vec_to_codes(self.encoder(self.preprocess_batch(images)))

    @torch.no_grad()
    def quantize(self, images: torch.Tensor) -> torch.Tensor:
        return self.quantizer.codes_to_vec(self.get_tokens(images))

    @torch.no_grad()
    def reconstruct(self, images: torch.Tensor) -> torch.Tensor:
        return self.preprocess_visualization(self(self.preprocess_batch(images))[0].float())

    @torch.no_grad()
    def reconstruct_from_tokens(self, tokens: torch.Tensor) -> torch.Tensor:
        b, s = tokens.shape
        side = int(round(s ** 0.5))
        q = self.quantizer.codes_to_vec(tokens).view(b, side, side, self.latent_dim).permute(0, 3, 1, 2)
        return self.preprocess_visualization(self.decoder(q).float())


class MSELoss(nn.Module):
    """``torch.nn.MSELoss`` stand-in on the HIP reduction kernel (vqvae/model.py:137)."""

    def forward(self, recon, target):
        return ops.mse_loss(recon.contiguous(memory_format=torch.channels_last), target)

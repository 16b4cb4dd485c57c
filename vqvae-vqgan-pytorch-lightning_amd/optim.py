"""Flat-arena AdamW: every trainable tensor is a view into ONE fp32 parameter buffer, ONE gradient buffer
and ONE second-moment buffer, updated by a single HIP launch (``vqk_adamw``) and all-reduced with a single
collective.  Semantics = ``torch.optim.AdamW`` as the reference configures it (vqvae/model.py:419-428:
two groups, weight decay only on conv weights; betas (0, 0.99) => the first moment is the gradient itself
and is not stored)."""
from __future__ import annotations

import torch
import torch.distributed as dist

from . import _native

_ALIGN = 64   # elements; keeps every tensor (fp32 and its bf16 shadow) 16-byte aligned inside the arena


def all_reduce_sum(t: torch.Tensor) -> None:
    """SUM all-reduce of ``t`` in place, ordered after the work queued on the current stream; the current stream waits for it.

    Issued as an ASYNC collective + ``wait()`` (a stream-side wait, the host does not block): the process group then runs the
    kernel on its own stream and records its completion events THERE.  A synchronous ``dist.all_reduce`` records them on the
    CURRENT stream -- and when that stream is the one a hipGraph is captured on a moment later (MiniTrainer's settling steps run on
    the capture stream), the group's watchdog thread polls an event 'last recorded in a capturing stream': hipErrorCapturedEvent,
    the watchdog terminates the process (seen 1 run in 3 with the full-size VQ-GAN step, where a capture takes seconds)."""
    from . import ops
    with ops.trace_range('vqk::all_reduce_sum'):
        work = _track(dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True))
        if work is not None:
            work.wait()


_PENDING: list = []          # Work handles of the collectives issued through this module's helpers since the last quiesce


def _track(work):
    if work is not None:
        _PENDING.append(work)
        del _PENDING[:-64]                                   # (completed long ago: only the recent ones can still be in flight)
    return work


def broadcast_(t: torch.Tensor, src: int = 0) -> None:
    """``dist.broadcast`` as async + stream-side wait (see :func:`all_reduce_sum`: no event is recorded on the CURRENT stream)"""
    work = _track(dist.broadcast(t, src=src, async_op=True))
    if work is not None:
        work.wait()


def barrier() -> None:
    """``dist.barrier`` as async + wait, for the same reason"""
    work = _track(dist.barrier(async_op=True))
    if work is not None:
        work.wait()


def quiesce_collectives(timeout_s: float = 30.0) -> None:
    """before a hipGraph capture: every collective issued through the helpers above has COMPLETED (polled on its Work handle, not
    assumed from a sleep) -- with async collectives the process group records its events on its own stream, so its watchdog never
    meets an event 'last recorded in a capturing stream'"""
    import time
    torch.cuda.synchronize()
    t0 = time.time()
    for w in _PENDING:
        while not w.is_completed():
            if time.time() - t0 > timeout_s:
                raise RuntimeError('vqk: a collective issued before the capture has not completed')
            time.sleep(0.001)
    _PENDING.clear()


def _is_channels_last_param(p: torch.Tensor) -> bool:
    return p.dim() == 4 and not p.is_contiguous() and p.is_contiguous(memory_format=torch.channels_last)


class FlatAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, arena_front=None, arena_back=None):
        """``arena_front``: ids of the parameters laid out FIRST in the arenas (any order of ``param_groups`` -- which is what
        ``state_dict`` is indexed by -- is kept): ``front_numel`` elements that can be all-reduced on their own;
        ``arena_back``: ids laid out LAST, from ``back_start`` on (the tensors whose gradients are ready last)."""
        defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        plist = [(p, g['weight_decay']) for g in self.param_groups for p in g['params']]
        if not plist:
            raise ValueError('FlatAdamW got no parameters')
        dev = plist[0][0].device
        front, back = arena_front or set(), arena_back or set()
        order = sorted(range(len(plist)), key=lambda i: (0 if id(plist[i][0]) in front else 2 if id(plist[i][0]) in back else 1, i))
        offs, total = [0] * len(plist), 0
        self.front_numel = 0
        self.back_start = None                # offset of the first ``arena_back`` tensor (== total when there is none)
        for i in order:
            p = plist[i][0]
            if p.dtype != torch.float32 or p.device != dev:
                raise ValueError('FlatAdamW needs fp32 parameters on one device')
            if id(p) in back and id(p) not in front and self.back_start is None:
                self.back_start = total
            offs[i] = total
            total += -(-p.numel() // _ALIGN) * _ALIGN
            if id(p) in front:
                self.front_numel = total
        if self.back_start is None:
            self.back_start = total
        self.flat_p = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_v = torch.zeros(total, dtype=torch.float32, device=dev)
        b1 = self.defaults['betas'][0]
        self.flat_m = torch.zeros(total, dtype=torch.float32, device=dev) if b1 != 0.0 else None
        seg_end, seg_wd = [], []
        with torch.no_grad():
            for (p, wd), off in ((plist[i], offs[i]) for i in order):      # segments in ARENA order (binary-searched by the kernel)
                n = p.numel()
                for buf, is_grad in ((self.flat_p, False), (self.flat_g, True)):
                    seg = buf[off:off + n]
                    if _is_channels_last_param(p):
                        o, i, kh, kw = p.shape
                        view = seg.view(o, kh, kw, i).permute(0, 3, 1, 2)
                    else:
                        view = seg.view(p.shape)
                    if is_grad:
                        p.grad = view
                        p._vqk_direct_grad = True     # ops.* may accumulate into the arena and skip AccumulateGrad
                        p._vqk_owner = self           # packed conv operands are refreshed by this optimizer's step()
                    else:
                        view.copy_(p.data)
                        p.data = view
                seg_end.append(off + n)
                seg_wd.append(float(wd))
                pad_end = off + -(-n // _ALIGN) * _ALIGN
                if pad_end != off + n:
                    seg_end.append(pad_end)
                    seg_wd.append(0.0)
        self.seg_end = torch.tensor(seg_end, dtype=torch.int64, device=dev)
        self.seg_wd = torch.tensor(seg_wd, dtype=torch.float32, device=dev)
        self.offsets = {id(p): off for (p, _), off in zip(plist, offs)}
        self.step_count = 0
        self.force_collective = False         # issue the all-reduce even at world size 1 (single-GPU RCCL smoke test)
        self.grad_scale = 1.0
        self.shadow = None                    # optional bf16 copy of flat_p refreshed by step()
        self.generation = 0                   # bumped by every step(): cache key for packed weights

    def enable_bf16_shadow(self):
        if self.shadow is None:
            self.shadow = self.flat_p.to(torch.bfloat16)
        return self.shadow

    # ---- checkpoint format: the same layout ``torch.optim.AdamW.state_dict()`` produces (what Lightning stores under
    # 'optimizer_states', vqvae/train.py:121-122), so optimizer states move between the reference and this build
    def _params_in_order(self):
        return [p for g in self.param_groups for p in g['params']]

    def state_dict(self):
        state, idx = {}, 0
        groups = []
        for g in self.param_groups:
            ids = []
            for p in g['params']:
                off, n = self.offsets[id(p)], p.numel()
                entry = {'step': torch.tensor(float(self.step_count)),
                         'exp_avg_sq': self._logical(self.flat_v[off:off + n], p).clone()}
                entry['exp_avg'] = (self._logical(self.flat_m[off:off + n], p).clone() if self.flat_m is not None
                                    else torch.zeros_like(p, memory_format=torch.contiguous_format))
                state[idx] = entry
                ids.append(idx)
                idx += 1
            groups.append({**{k: v for k, v in g.items() if k != 'params'}, 'params': ids})
        return {'state': state, 'param_groups': groups}

    @staticmethod
    def _logical(seg, p):
        """arena segment (conv weights: [O][kh][kw][I] memory) -> contiguous tensor of the parameter's logical shape"""
        if _is_channels_last_param(p):
            o, i, kh, kw = p.shape
            return seg.view(o, kh, kw, i).permute(0, 3, 1, 2).contiguous()
        return seg.view(p.shape)

    @torch.no_grad()
    def load_state_dict(self, sd):
        params = self._params_in_order()
        # torch.optim.AdamW keeps state only for parameters that received a gradient (the reference's frozen EMA codebook
        # never does): the state may be SPARSE; its keys index the parameter list, which 'param_groups' pins
        saved = sum(len(g['params']) for g in sd['param_groups'])
        if saved != len(params) or any(int(i) >= len(params) for i in sd['state']):
            raise ValueError(f'FlatAdamW.load_state_dict: the saved optimizer holds {saved} parameters, this one {len(params)} '
                             f'(reference checkpoints need optimizer_param_set="reference")')
        for g, gs in zip(self.param_groups, sd['param_groups']):
            if len(g['params']) != len(gs['params']):
                raise ValueError('FlatAdamW.load_state_dict: parameter groups differ in size')
        for g, gs in zip(self.param_groups, sd['param_groups']):
            for k, v in gs.items():
                if k != 'params':
                    g[k] = tuple(v) if k == 'betas' else v
        for idx, p in enumerate(params):
            st = sd['state'].get(idx, sd['state'].get(str(idx)))
            if st is None:
                continue
            off, n = self.offsets[id(p)], p.numel()
            for name, buf in (('exp_avg_sq', self.flat_v), ('exp_avg', self.flat_m)):
                if buf is None or name not in st:
                    continue
                src = st[name].to(buf.device, torch.float32)
                if tuple(src.shape) != tuple(p.shape):
                    raise ValueError(f'FlatAdamW.load_state_dict: state {idx} ({name}) has shape {tuple(src.shape)}, '
                                     f'parameter {idx} has {tuple(p.shape)} -- the parameter ORDER of the saved optimizer '
                                     f'differs (reference checkpoints need optimizer_param_set="reference")')
                if _is_channels_last_param(p):
                    src = src.permute(0, 2, 3, 1)
                buf[off:off + n].copy_(src.reshape(-1))
            self.step_count = int(float(st['step']))

    def zero_grad(self, set_to_none: bool = False):
        self.flat_g.zero_()

    def collective_on(self) -> bool:
        return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or self.force_collective)

    # bench.py: collectives issued / bytes handed to RCCL since construction, and a timing-only switch that skips the
    # collectives (the step then runs exactly as at world 1: step time with - without = the EXPOSED communication time)
    collectives_issued = 0
    collective_bytes = 0
    mute_collectives = False

    def _count(self, numel: int) -> None:
        self.collectives_issued += 1
        self.collective_bytes += numel * 4

    def all_reduce_grads(self, world_size: int | None = None):
        """ONE collective per optimizer step (replaces DDP's bucketed reducer, vqvae/train.py:128)."""
        if self.collective_on():
            if not self.mute_collectives:
                all_reduce_sum(self.flat_g)
                self._count(self.flat_g.numel())
            self.grad_scale = 1.0 / dist.get_world_size()
        else:
            self.grad_scale = 1.0

    def all_reduce_range(self, lo: int, hi: int, async_op: bool = True):
        """all-reduce of arena elements [lo, hi) (the overlapped form: the front range -- the decoder's gradients -- while the
        rest of the backward still runs; RCCL orders it after the work already queued on the current stream).  Returns the
        work handle (None when no collective is needed)."""
        if not self.collective_on() or hi <= lo:
            self.grad_scale = 1.0 / dist.get_world_size() if self.collective_on() else 1.0
            return None
        self.grad_scale = 1.0 / dist.get_world_size()
        if self.mute_collectives:
            return None
        self._count(hi - lo)
        work = dist.all_reduce(self.flat_g[lo:hi], op=dist.ReduceOp.SUM, async_op=async_op)
        return _track(work) if async_op else work

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lrs = {g['lr'] for g in self.param_groups}
        if len(lrs) != 1:
            raise RuntimeError('FlatAdamW: all param groups must share one learning rate (as the reference sets them)')
        g0 = self.param_groups[0]
        b1, b2 = g0['betas']
        self.step_count += 1
        if not self.flat_p.is_cuda:
            raise RuntimeError('vqk: FlatAdamW.step runs on the GPU only (HIP kernel, no CPU fallback)')
        from . import ops
        with ops.trace_range('vqk::FlatAdamW.step'):
            st = self._launch_adamw(g0, b1, b2)
        _native.check(st, 'adamw')
        self.generation += 1
        with ops.trace_range('vqk::repack_owned'):
            ops.repack_owned(self)               # every cached conv operand of these weights, one launch
        return loss

    def _launch_adamw(self, g0, b1, b2):
        return _native.lib().vqk_adamw(self.flat_p.data_ptr(), self.flat_g.data_ptr(),
                                     0 if self.flat_m is None else self.flat_m.data_ptr(), self.flat_v.data_ptr(),
                                     self.flat_p.numel(), self.seg_end.data_ptr(), self.seg_wd.data_ptr(),
                                     self.seg_end.numel(), float(g0['lr']), float(b1), float(b2), float(g0['eps']),
                                     self.step_count, float(self.grad_scale),
                                     0 if self.shadow is None else self.shadow.data_ptr(),
                                     torch.cuda.current_stream().cuda_stream)

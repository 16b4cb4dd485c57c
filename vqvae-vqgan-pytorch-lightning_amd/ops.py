"""Host-side operators over the vqk C-ABI: thin launchers (``raw_*``) and the ``torch.autograd.Function``s
built from them.  PyTorch is used for device memory, streams and the autograd *graph*; every
arithmetic kernel on the path is a hand-written gfx950 kernel reached through ``libvqk.so``.

Tensor convention: activations keep the reference's logical NCHW shape but are physically NHWC
(``torch.channels_last``), conv weights keep logical OIHW and are physically [O][kh][kw][I].
"""
from __future__ import annotations

import os
import threading
import weakref

import torch

from . import _native

F32, BF16 = 0, 1
_CL = torch.channels_last


def dcode(dtype: torch.dtype) -> int:
    if dtype == torch.float32:
        return F32
    if dtype == torch.bfloat16:
        return BF16
    raise RuntimeError(f'vqk: unsupported dtype {dtype}')


def epc(dtype: torch.dtype) -> int:
    """elements per 16-byte chunk: channel counts handed to the conv kernels are multiples of this"""
    return 4 if dtype == torch.float32 else 8


DETERMINISTIC = False
_DET_WS: dict = {}
_DET_TLS = threading.local()           # the library's flag is per HOST THREAD (autograd runs the backward on its own thread):
_DET_GEN = 0                           # every thread tracks what IT armed, against the generation of the last mode switch
_DET_WS_BYTES = 64 << 20
_SCRATCH: dict = {}
_SCRATCH_BYTES = 64 << 20
_TILE_QUEUE: dict = {}               # _wkey() -> int32 words of the conv kernels' dynamic tile queue (include/vqk.h)


def _wkey(device=None) -> tuple:
    """key of every workspace a kernel WRITES: (device index, stream handle, host thread).  Per stream: launches on two streams may
    run concurrently.  Per host thread: two Python threads that share a stream (the default one, typically) enqueue in an arbitrary
    interleaving -- a producer's sums / split-K slices must not meet the other thread's between two launches
    (tests/test_gpu_two_models.py)."""
    dev = torch.cuda.current_device() if device is None else (device.index if device.index is not None else torch.cuda.current_device())
    return (dev, torch.cuda.current_stream().cuda_stream, threading.get_ident())


class _WsCtx:
    """one ``vqk_ctx`` (include/vqk.h) = the workspaces of ONE (device, stream, host thread): split-K scratch, tile-queue words and --
    armed lazily, when deterministic mode is on -- the ordered-sum slices.  Created once, named with ONE call when the thread's
    current (device, stream) changes (rounds 1-5 re-armed three thread-local pointers instead)."""
    __slots__ = ('handle', 'scratch', 'tq', 'det', 'det_gen')

    def __init__(self, dev: int):
        import ctypes
        lib = _native.lib()
        h = ctypes.c_void_p()
        _native.check(lib.vqk_ctx_create(ctypes.byref(h)), 'ctx_create')
        self.handle = h.value
        self.scratch = torch.empty(_SCRATCH_BYTES // 4, dtype=torch.float32, device=f'cuda:{dev}')      # no zero fill: every slice is written before it is summed
        self.tq = torch.zeros(64, dtype=torch.int32, device=f'cuda:{dev}')                              # zero on entry, left zero by the kernels
        _native.check(lib.vqk_ctx_set_scratch(self.handle, self.scratch.data_ptr(), self.scratch.numel() * 4), 'ctx_set_scratch')
        _native.check(lib.vqk_ctx_set_tile_queue(self.handle, self.tq.data_ptr(), self.tq.numel() * 4), 'ctx_set_tile_queue')
        self.det, self.det_gen = None, -1


_WS_CTX: dict = {}                   # _wkey() -> _WsCtx


def _stream() -> int:
    """handle of the current stream (every launcher passes it to the C-ABI) -- and the point where the library's workspace context
    follows the calling thread's (device, stream): partial sums of two concurrent streams, or of two host threads that share a
    stream, must not share a buffer.  Deterministic mode arms / disarms the context's ordered-sum workspace here."""
    s = torch.cuda.current_stream().cuda_stream
    dev = torch.cuda.current_device()                            # (the default stream's handle is 0 on every device)
    cur = getattr(_DET_TLS, 'ctx', None)
    if cur is None or cur[0] != (dev, s):
        wk = (dev, s, threading.get_ident())
        ctx = _WS_CTX.get(wk)
        if ctx is None:
            ctx = _WS_CTX[wk] = _WsCtx(dev)
            _SCRATCH[wk], _TILE_QUEUE[wk] = ctx.scratch, ctx.tq                  # (names the tests and tools read)
        _native.check(_native.lib().vqk_ctx_make_current(ctx.handle), 'ctx_make_current')
        _DET_TLS.ctx = cur = ((dev, s), ctx)
    ctx = cur[1]
    if DETERMINISTIC:
        if ctx.det_gen != _DET_GEN:
            if ctx.det is None:
                ctx.det = torch.empty(_DET_WS_BYTES, dtype=torch.uint8, device=f'cuda:{dev}')
                _DET_WS[(dev, s, threading.get_ident())] = ctx.det
            _native.check(_native.lib().vqk_ctx_set_deterministic(ctx.handle, 1, ctx.det.data_ptr(), ctx.det.numel()), 'ctx_set_deterministic')
            ctx.det_gen = _DET_GEN
    elif ctx.det_gen != -1:
        _native.check(_native.lib().vqk_ctx_set_deterministic(ctx.handle, 0, 0, 0), 'ctx_set_deterministic')
        ctx.det_gen = -1
    return s


def rearm_workspaces() -> None:
    """after a caller went around this module and changed the calling thread's CURRENT context through the C-ABI setters
    (vqk_set_scratch / vqk_set_tile_queue / vqk_set_deterministic -- tests do): put this thread's context back in order"""
    cur = getattr(_DET_TLS, 'ctx', None)
    if cur is not None:
        ctx, lib = cur[1], _native.lib()
        _native.check(lib.vqk_ctx_set_scratch(ctx.handle, ctx.scratch.data_ptr(), ctx.scratch.numel() * 4), 'ctx_set_scratch')
        _native.check(lib.vqk_ctx_set_tile_queue(ctx.handle, ctx.tq.data_ptr(), ctx.tq.numel() * 4), 'ctx_set_tile_queue')
        _native.check(lib.vqk_ctx_set_deterministic(ctx.handle, 0, 0, 0), 'ctx_set_deterministic')
        ctx.det_gen = -1
    _DET_TLS.ctx = None


X3 = _native.switch('VQK_CONV_PRODUCTS', 'fp32') == 'bf16x3'
X3_WGRAD_FOLD = _native.switch('VQK_X3_WGRAD_FOLD', '0') == '1'     # A/B: the pair-tensor form (two split passes + the folded bf16 role-split kernel) instead of conv3x3_wgrad_x3_kernel


def set_conv_products(mode: str) -> None:
    """How the fp32 compute mode multiplies in the 3x3 convs: 'fp32' -- exact v_mfma_f32_32x32x2_f32 (the reference mode: what
    the CPU oracle computes, bit for bit per product); 'bf16x3' -- every product as three bf16 products with fp32 accumulation
    (csrc/conv_x3.hip: x_hi w_hi + x_lo w_hi + x_hi w_lo, ~2^-17 relative per product) on the bf16 matrix pipe, fp32 activation
    storage, GroupNorm / quantizer / optimizer unchanged.  Read at FORWARD time (the autograd nodes carry it to their backward)."""
    global X3
    if mode not in ('fp32', 'bf16x3'):
        raise ValueError("conv products: 'fp32' or 'bf16x3'")
    X3 = mode == 'bf16x3'


def x3_serves(dtype, out_dtype, h_out: int, w_out: int, cin: int, cout: int, ksize: int) -> bool:
    """the split-product 3x3 kernel serves this problem (fp32 in / out, whole 32-channel chunks, 8x16-pixel tiles)"""
    return (dtype == torch.float32 and out_dtype in (None, torch.float32) and ksize in (1, 3) and cin % 32 == 0 and cout % 32 == 0
            and h_out % 8 == 0 and w_out % 16 == 0)


def check_kernel_health() -> None:
    """Raise if a kernel of the run reported that it gave up on a cross-block rendezvous (the cluster GroupNorm backward's bounded
    spin, csrc/norm.hip: a block that times out continues with INCOMPLETE sums -- a wrong gradient, not a hang).  Synchronises the
    device: called where a run synchronises anyway (end of an epoch, before a checkpoint is written, end of the benchmark's timed
    region), never inside a step."""
    import ctypes
    cnt = ctypes.c_int(0)
    _native.check(_native.lib().vqk_gn_cluster_timeouts(ctypes.byref(cnt)), 'gn_cluster_timeouts')
    if cnt.value:
        raise RuntimeError(f'vqk: {cnt.value} GroupNorm cluster block(s) timed out waiting for their partners: the gradients of this run '
                           'are not trustworthy (set the tuning slot GN_CLUSTER_MAX_HW to 0 for the two-kernel passes)')


def set_deterministic(on: bool) -> None:
    """``pl.Trainer(deterministic=True)`` (vqvae/train.py:130) for the vqk kernels: ordered partial sums instead of atomics in
    arrival order (include/vqk.h: vqk_set_deterministic); the GroupNorm sums are no longer fused into the conv drains (those
    use atomics).  Two identical steps then produce bit-identical gradients."""
    global DETERMINISTIC, _DET_GEN, FUSE_GN_STATS
    DETERMINISTIC = bool(on)
    _DET_GEN += 1
    _DB_DONE.clear()
    FUSE_GN_STATS = _native.switch('VQK_FUSE_GN_STATS', '1') != '0'     # (deterministic mode: per-tile slots instead of atomics)
    _GN_WS.clear(); _GN_PARTS.clear()


def _p(t):
    return 0 if t is None else t.data_ptr()


def _require_gpu(t: torch.Tensor):
    if not t.is_cuda:
        raise RuntimeError('vqk: operators run on the GPU only (HIP kernels, no CPU fallback); got a CPU tensor')


_zero_pages = {}


def zero_page(device) -> torch.Tensor:
    key = (device.type, device.index)
    if key not in _zero_pages:
        _zero_pages[key] = torch.zeros(1024, dtype=torch.uint8, device=device)
    return _zero_pages[key]


def nhwc(x: torch.Tensor) -> torch.Tensor:
    return x.contiguous(memory_format=_CL)


def empty_nhwc(n, c, h, w, dtype, device) -> torch.Tensor:
    return torch.empty((n, c, h, w), dtype=dtype, device=device, memory_format=_CL)


# ------------------------------------------------------------------------------------------------------
# raw launchers
# ------------------------------------------------------------------------------------------------------
KERNEL_EVENTS = None      # bench.py sets this to a list: (kernel, algorithmic FLOPs, algorithmic bytes, start, stop)


_EVENT_SHAPES = _native.switch('VQK_EVENT_SHAPES', '') == '1'     # tooling: one statistics line per (kernel, FLOP count)


def _timed(name: str, flops: float, launch, nbytes: float = 0.0, launches: int = 1, exec_flops: float | None = None):
    """``launches``: kernel launches behind this one event (an upsample conv in phase form is four); ``exec_flops``: the
    multiply-adds the launches actually execute when that differs from the algorithmic count (phase form: 4/9)"""
    if KERNEL_EVENTS is None:
        return launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    st = launch()
    e1.record()
    KERNEL_EVENTS.append((name, flops, nbytes, e0, e1, launches, flops if exec_flops is None else exec_flops))
    return st


_MX_ON = _native.switch('VQK_MX', '1') != '0'
_THIN_OUT = _native.switch('VQK_THIN_OUT', '1') != '0'
_MX_MIN_TILES = int(_native.switch('VQK_MX_MIN_TILES', '1'))
_WGMX_ON = _native.switch('VQK_WGMX', '1') != '0'
_UPS_MERGE = _native.switch('VQK_UPS_MERGE', '1') != '0'      # (the library's tuning slot of the same name: one launch per phase-form conv)


def _fprop_kernel_name(dtype, wlayout: int, shape=None) -> str:
    """the kernel symbol the launcher picks (csrc/conv.hip::launch_fprop), for the per-kernel event statistics;
    shape = (n, h_out, w_out, cin, cout, act, out_dtype) lets it tell the matrix/auxiliary-wave kernel from the stream kernel"""
    name = _fprop_kernel_name0(dtype, wlayout, shape)
    if _EVENT_SHAPES and shape is not None:                      # tooling (tools/per_shape.py): one line per layer shape
        name += f' {shape[3]}->{shape[4]}@{shape[1]}x{shape[2]}'
    return name


def _fprop_kernel_name0(dtype, wlayout: int, shape=None) -> str:
    if wlayout == 5:
        return 'conv3x3_x3_kernel<f32 as 3 x bf16>'
    if wlayout == 1:
        if dtype != torch.bfloat16:
            return 'conv3x3_halo_breg_kernel<f32>'
        if shape is not None and _MX_ON:
            n, ho, wo, cin, cout, act, odt = shape
            if (act in (0, 2, 3) and odt == torch.bfloat16 and cout % 128 == 0 and cin >= 64
                    and n * ho * wo // 256 * (cout // 128) >= _MX_MIN_TILES):
                return 'conv3x3_mx_kernel<bf16>'
        return 'conv3x3_stream_kernel<bf16>'
    return f'conv_fprop_kernel<{"f32" if dtype == torch.float32 else "bf16"}>'



def weight_layout(dtype, n, h_in, w_in, cin, cout, ksize, ups, out_dtype=None, x3=None) -> int:
    """operand layout the fprop launcher wants for this problem (include/vqk.h: 0 = [O][kh][kw][I], 1 = fragment-major,
    5 = fragment-major (hi | lo) bf16 pairs of the split-product mode; ``x3``: that mode on / off, None = the process setting);
    a 1x1 conv has a fragment-major form only on the bf16 -> bf16 matrix/auxiliary-wave kernel"""
    if (X3 if x3 is None else x3) and not (ups and ksize == 1) and x3_serves(dtype, out_dtype, h_in << int(ups), w_in << int(ups), cin, cout, ksize):
        return 5                                   # split products on the bf16 matrix pipe (csrc/conv_x3.hip)
    if ksize == 1 and (dtype != torch.bfloat16 or (out_dtype is not None and out_dtype != torch.bfloat16)):
        return 0
    if (_THIN_OUT and ksize == 3 and not ups and dtype == torch.bfloat16 and out_dtype in (None, torch.bfloat16)
            and cin == 128 and cout == 8 and h_in % 8 == 0 and w_in % 32 == 0):
        return 0                                   # the decoder's last conv: plain weights for vqk_conv2d_thin_out
    r = _native.lib().vqk_conv_weight_layout(dcode(dtype), n, h_in, w_in, cin, cout, ksize, int(ups))
    if r < 0:
        _native.check(r, 'conv_weight_layout')
    return r


def pack_weights(w_mem_f32, dtype, cout, cin, ksize, transpose: bool, layout: int) -> torch.Tensor:
    """fp32 [Cout][k][k][Cin] master memory -> conv operand (cast / transpose+flip for dgrad / fragment-major)"""
    if layout == 0 and not transpose and dtype == torch.float32:
        return w_mem_f32
    dc, di = (cin, cout) if transpose else (cout, cin)
    n = _native.lib().vqk_conv_packed_elems(dc, di, ksize, layout)
    out = torch.empty(n, dtype=dtype, device=w_mem_f32.device)
    st = _native.lib().vqk_conv_pack_weights(w_mem_f32.data_ptr(), out.data_ptr(), dcode(dtype), cout, cin, ksize,
                                             int(transpose), layout, _stream())
    _native.check(st, 'conv_pack_weights')
    return out


class _PackEntry:
    __slots__ = ('wref', 'src', 'dst', 'desc', 'stamp', 'ready', 'stage')


class _PackStage:
    """persistent fp32 image of a weight in the kernels' [O_pad][k][k][I_pad] order, for weights whose own memory is not that:
    zero-padded channel counts (the 3-channel edge convs, the discriminator's fromrgb), torch-contiguous OIHW parameters (VGG16,
    the StyleGAN2 layers), 2-D fully connected weights viewed as 1x1 convs, 1x1 weights placed at the centre tap of a 3x3
    (``kind = 'centre3'``).  Refreshed by ONE strided copy when the master weight changes -- together with the packed operands,
    i.e. after the optimizer step (:func:`repack_owned`), never inside the captured step.  (Before round 5 these weights were
    zero-filled, copied and packed per conv CALL: 9 launches per headline step, 60 per VQ-GAN step.)"""
    __slots__ = ('wref', 'buf', 'stamp', 'shape4', 'kind')

    def refresh(self, weight, stamp) -> None:
        if self.stamp == stamp:
            return
        o, i, k, _ = self.shape4
        w4 = weight.detach().reshape(o, i, k, k)
        with torch.no_grad():
            if self.kind == 'centre3':
                self.buf[:o, 1, 1, :i].copy_(w4.reshape(o, i))
            else:
                self.buf[:o, :, :, :i].copy_(w4.permute(0, 2, 3, 1))
        self.stamp = stamp


_PACK_STAGES: dict = {}         # (data_ptr, shape4, cin_pad, cout_pad, kind) -> _PackStage


def _pack_stage(weight, shape4, cin_pad: int, cout_pad: int, kind: str) -> _PackStage:
    key = (weight.data_ptr(), tuple(shape4), cin_pad, cout_pad, kind)
    st = _PACK_STAGES.get(key)
    if st is not None and st.wref() is not weight:
        st = None
    if st is None:
        st = _PackStage()
        k = 3 if kind == 'centre3' else shape4[2]
        st.wref, st.shape4, st.kind, st.stamp = weakref.ref(weight), tuple(shape4), kind, None
        st.buf = torch.zeros((cout_pad, k, k, cin_pad), dtype=torch.float32, device=weight.device)
        _PACK_STAGES[key] = st
    return st


def _pack_written(entries) -> None:
    """``entries`` were (re)written by a launch on the current stream: a use from ANOTHER stream has to wait for it.  (The VQ-GAN
    step runs LPIPS and the discriminator's real pass on second streams; a backward operand packed lazily by one of two
    concurrent chains was read by the other while the pack kernel was still running -- one golden-step failure in twelve runs.)"""
    if torch.cuda.is_current_stream_capturing():
        ready = None                                     # (captured work is ordered by the capture; settled steps pack nothing lazily)
    else:
        ev = torch.cuda.Event()
        ev.record()
        ready = (torch.cuda.current_stream(), ev)
    for ent in entries:
        ent.ready = ready


def _pack_use(ent) -> torch.Tensor:
    r = ent.ready
    if r is not None:
        cur = torch.cuda.current_stream()
        if cur != r[0] and not torch.cuda.is_current_stream_capturing():
            cur.wait_event(r[1])
    return ent.dst


_PACK_CACHE: dict = {}          # (data_ptr, shape4, k, transpose, layout, dtype, cin_pad, cout_pad, kind) -> _PackEntry
_PACK_LOCK = threading.RLock()  # cache bookkeeping (the launches themselves are ordered by their streams)
_PACK_BLOCKS = int(_native.switch('VQK_PACK_BLOCKS', '128'))    # blocks per operand of the repack launch (32 -> 128: -0.05 ms/step, the 512x512x9 operands)
_PACK_TABLE = None              # {tuple of keys: device int64 [n, 8]} descriptor tables of the repack launches


def _pack_stamp(weight):
    """(in-place version of the parameter, generation of the FlatAdamW that owns it): changes whenever the master
    weights change, through torch (``_version``) or through the AdamW kernel (``generation``)."""
    owner = getattr(weight, '_vqk_owner', None)
    return (weight._version, owner.generation if owner is not None else -1)


def packed_weight(weight, cin_pad: int, cout_pad: int, dtype, ksize: int, transpose: bool, layout: int, shape4=None,
                  kind: str = 'pad') -> torch.Tensor:
    """The conv operand of ``weight`` (logical [O,I,k,k]; ``shape4``: that shape for a 2-D fully connected weight) in ``dtype`` /
    ``layout``, from a cache of persistent buffers that is refreshed by ONE multi-tensor launch per optimizer step
    (:func:`repack_owned`) instead of one pack launch per conv call.  ``kind = 'centre3'``: a 1x1 weight as the centre tap of a
    3x3 operand (ksize = 3)."""
    w = weight.detach()
    shape4 = tuple(shape4) if shape4 is not None else tuple(w.shape)
    o, i = shape4[0], shape4[1]
    direct = (kind == 'pad' and cin_pad == i and cout_pad == o and w.dim() == 4 and w.permute(0, 2, 3, 1).is_contiguous())
    if direct and layout == 0 and not transpose and dtype == torch.float32:
        return w.permute(0, 2, 3, 1).reshape(-1)
    key = (w.data_ptr(), shape4, ksize, bool(transpose), layout, dtype, cin_pad, cout_pad, kind)
    with _PACK_LOCK:
        return _packed_weight_locked(weight, w, key, direct, cin_pad, cout_pad, dtype, ksize, transpose, layout, shape4, kind)


def _packed_weight_locked(weight, w, key, direct, cin_pad, cout_pad, dtype, ksize, transpose, layout, shape4, kind):
    ent = _PACK_CACHE.get(key)
    stamp = _pack_stamp(weight)
    if ent is not None and ent.wref() is not weight:
        ent = None                                       # the address was recycled by another parameter
    if ent is not None and ent.stamp == stamp:
        return _pack_use(ent)
    if ent is None:
        dc, di = (cin_pad, cout_pad) if transpose else (cout_pad, cin_pad)
        ent = _PackEntry()
        ent.wref = weakref.ref(weight)
        ent.stage = None if direct else _pack_stage(weight, shape4, cin_pad, cout_pad, kind)
        ent.src = w.permute(0, 2, 3, 1).reshape(-1) if direct else ent.stage.buf.reshape(-1)
        if layout == 0 and not transpose and dtype == torch.float32:
            ent.dst = ent.src                            # the padded fp32 image IS the operand
        else:
            ent.dst = torch.empty(_native.lib().vqk_conv_packed_elems(dc, di, ksize, layout), dtype=dtype, device=w.device)
        ent.desc = [ent.src.data_ptr(), ent.dst.data_ptr(), dcode(dtype), cout_pad, cin_pad, ksize, int(transpose), layout]
        ent.ready = None
        _PACK_CACHE[key] = ent
    if ent.stage is not None:
        ent.stage.refresh(weight, stamp)
    if ent.dst is not ent.src:
        st = _native.lib().vqk_conv_pack_weights(ent.src.data_ptr(), ent.dst.data_ptr(), dcode(dtype), cout_pad, cin_pad, ksize,
                                                 int(transpose), layout, _stream())
        _native.check(st, 'conv_pack_weights')
    ent.stamp = stamp
    _pack_written((ent,))
    return ent.dst


def repack_owned(owner=None) -> int:
    """Refresh every cached operand whose master weight belongs to ``owner`` (a FlatAdamW; None: every stale entry)
    with one ``vqk_conv_pack_multi`` launch.  Called by ``FlatAdamW.step`` right after the AdamW kernel."""
    global _PACK_TABLE
    refresh_vq_prepared(owner)                           # the quantizer's prepared codebook follows the same rule
    refresh_padded_vectors(owner)
    with _PACK_LOCK:                                     # (two models stepped from two host threads share the cache)
        ents = []
        for key, ent in list(_PACK_CACHE.items()):
            weight = ent.wref()
            if weight is None or weight.data_ptr() != key[0]:
                _PACK_CACHE.pop(key, None)               # parameter freed or re-pointed (e.g. into a flat arena)
            elif owner is not None:
                if getattr(weight, '_vqk_owner', None) is owner:
                    ents.append(ent)
            elif ent.stamp != _pack_stamp(weight):
                ents.append(ent)
        for key, st in list(_PACK_STAGES.items()):
            if st.wref() is None or st.wref().data_ptr() != key[0]:
                _PACK_STAGES.pop(key, None)
        if not ents:
            return 0
        for ent in ents:                                 # the padded / re-ordered fp32 images first (one strided copy each)
            if ent.stage is not None:
                ent.stage.refresh(ent.wref(), _pack_stamp(ent.wref()))
        packs = [ent for ent in ents if ent.dst is not ent.src]
        if packs:
            if _PACK_TABLE is None:
                _PACK_TABLE = {}
            # one device table per set of DESCRIPTORS (two optimizers alternate in the VQ-GAN step).  Keyed by the descriptors
            # themselves: a later model may get the same parameter addresses with other destination buffers
            sig = tuple(tuple(int(v) for v in ent.desc) for ent in packs)
            table = _PACK_TABLE.get(sig)
            if table is None:
                if len(_PACK_TABLE) >= 8:
                    _PACK_TABLE.clear()
                table = _PACK_TABLE[sig] = torch.tensor([list(d) for d in sig], dtype=torch.int64).to(packs[0].dst.device)
            _native.check(_native.lib().vqk_conv_pack_multi(table.data_ptr(), len(packs), _PACK_BLOCKS, _stream()), 'conv_pack_multi')
        for ent in ents:
            ent.stamp = _pack_stamp(ent.wref())
        _pack_written(ents)
        return len(ents)


class _PadVec:
    __slots__ = ('wref', 'buf', 'stamp')


_PAD_VECS: dict = {}            # (data_ptr, n_pad) -> _PadVec: a bias zero-padded to the kernels' channel count


def padded_vector(vec, n_pad: int) -> torch.Tensor:
    """fp32 copy of the 1-D parameter ``vec`` zero-padded to ``n_pad`` elements, persistent and refreshed like the packed
    operands (a conv whose output channels are padded -- the decoder's 3-channel head -- zero-filled and copied its bias per call)"""
    v = vec.detach()
    if v.numel() == n_pad and v.dtype == torch.float32:
        return v
    key = (v.data_ptr(), n_pad)
    ent = _PAD_VECS.get(key)
    if ent is not None and ent.wref() is not vec:
        ent = None
    if ent is None:
        ent = _PadVec()
        ent.wref, ent.stamp = weakref.ref(vec), None
        ent.buf = torch.zeros(n_pad, dtype=torch.float32, device=v.device)
        _PAD_VECS[key] = ent
    stamp = _pack_stamp(vec)
    if ent.stamp != stamp:
        with torch.no_grad():
            ent.buf[:v.numel()].copy_(v)
        ent.stamp = stamp
    return ent.buf


def refresh_padded_vectors(owner=None) -> None:
    for key, ent in list(_PAD_VECS.items()):
        vec = ent.wref()
        if vec is None or vec.data_ptr() != key[0]:
            del _PAD_VECS[key]
        elif (owner is None and ent.stamp != _pack_stamp(vec)) or (owner is not None and getattr(vec, '_vqk_owner', None) is owner):
            with torch.no_grad():
                ent.buf[:vec.numel()].copy_(vec.detach())
            ent.stamp = _pack_stamp(vec)


def clear_pack_cache():
    global _PACK_TABLE
    _PACK_CACHE.clear()
    _PACK_STAGES.clear()
    _PAD_VECS.clear()
    _VQ_PREP.clear()
    _PACK_TABLE = None


def raw_conv_fprop(x, wq, bias, residual, ksize: int, ups: bool, act: int, out_dtype, cout: int, wlayout: int = 0, out=None):
    """x [N,Cin,H,W] nhwc; wq: packed weights (pack_weights) in x.dtype with layout ``wlayout``.  ``out``: write here (a batch slice
    of a larger nhwc tensor: the half-batch pipeline of ResBlockFn.backward)."""
    _require_gpu(x)
    n, cin, h, w = x.shape
    s = 2 if ups else 1
    y = out if out is not None else empty_nhwc(n, cout, h * s, w * s, out_dtype, x.device)
    flops = 2.0 * n * h * s * w * s * cout * cin * ksize * ksize
    nbytes = (x.numel() * x.element_size() + y.numel() * y.element_size() * (2 if residual is not None else 1)
              + cout * cin * ksize * ksize * x.element_size())
    if (_THIN_OUT and x.dtype == torch.bfloat16 and out_dtype == torch.bfloat16 and ksize == 3 and not ups and wlayout == 0
            and residual is None and act in (0, 1) and cin == 128 and cout == 8 and h % 8 == 0 and w % 32 == 0):
        # the decoder's last conv: 8 output channels on 16x16x32 MFMAs (csrc/conv_edge.hip)
        st = _timed('conv3x3_thin_out_kernel<bf16> (HBM)' + (f' {cin}->{cout}@{h}x{w}' if _EVENT_SHAPES else ''), 0.0,
                    lambda: _native.lib().vqk_conv2d_thin_out(dcode(x.dtype), x.data_ptr(), wq.data_ptr(), _p(bias), y.data_ptr(),
                                                              n, h, w, cin, cout, act, zero_page(x.device).data_ptr(), _stream()),
                    x.numel() * 2 + y.numel() * 2)
        _native.check(st, 'conv2d_thin_out')
        return y
    kname = _fprop_kernel_name(x.dtype, wlayout, (n, h * s, w * s, cin, cout, act, out_dtype))
    if (x.dtype == torch.bfloat16 and out_dtype == torch.bfloat16 and wlayout == 0 and ksize == 3 and cin == 8 and residual is None
            and not ups and w % 32 == 0 and cout % 8 == 0):
        # one 16-byte chunk of input channels (the padded 3-channel image): csrc/conv.hip::launch_fprop takes the
        # thin-input kernel -- it writes 2 * Cout bytes per pixel at memory speed: an HBM line, not an MFMA one
        kname, flops = kname.replace('conv_fprop_kernel<bf16>', 'conv3x3_thin_in_kernel<bf16> (HBM)'), 0.0
        nbytes = x.numel() * 2 + y.numel() * 2
    if ksize == 1 and wlayout == 5:
        # the NTAP = 1 form of the split-product kernel: fp32 in + out at memory speed -- an HBM line like the bf16 1x1 form
        kname, flops = 'conv1x1_x3_kernel<f32 as 3 x bf16> (HBM)' + (f' {cin}->{cout}@{h}x{w}' if _EVENT_SHAPES else ''), 0.0
    if ksize == 1 and kname.startswith('conv3x3_mx_kernel'):
        # the NTAP = 1 instantiation (ResBlock shortcuts): memory-bound, reported with its bytes like the GroupNorm passes
        kname, flops = kname.replace('conv3x3_mx_kernel<bf16>', 'conv1x1_mx_kernel<bf16> (HBM)'), 0.0
    st = _timed(kname, flops,
                lambda: _native.lib().vqk_conv2d_fprop(dcode(x.dtype), x.data_ptr(), wq.data_ptr(), _p(bias),
                                                       _p(residual), y.data_ptr(), dcode(out_dtype), n, h, w, cin,
                                                       cout, ksize, int(ups), act, wlayout,
                                                       zero_page(x.device).data_ptr(), _stream()), nbytes,
                exec_flops=3.0 * flops if wlayout == 5 else None)
    _native.check(st, 'conv2d_fprop')
    return y


def can_pool_epilogue(dtype, cout: int, wlayout: int) -> bool:
    """the fused 2x2-pooling epilogue exists on the bf16 stream kernel for whole 128-wide cout tiles"""
    return dtype == torch.bfloat16 and wlayout == 1 and cout % 128 == 0


def raw_conv_fprop_pooled(x, wq, bias, residual, ksize: int, ups: bool, cout: int, pool_scale: float):
    """pool_scale * sum-pool2x2(conv(x) + bias + residual), written at half resolution (vqk_conv2d_fprop_pooled)"""
    _require_gpu(x)
    n, cin, h, w = x.shape
    s = 2 if ups else 1
    y = empty_nhwc(n, cout, h * s // 2, w * s // 2, x.dtype, x.device)
    flops = 2.0 * n * h * s * w * s * cout * cin * ksize * ksize
    nbytes = (x.numel() * x.element_size() + y.numel() * y.element_size()
              + (residual.numel() * residual.element_size() if residual is not None else 0)
              + cout * cin * ksize * ksize * x.element_size())
    st = _timed(_fprop_kernel_name(x.dtype, 1, (n, h * s, w * s, cin, cout, 0, x.dtype)), flops,
                lambda: _native.lib().vqk_conv2d_fprop_pooled(dcode(x.dtype), x.data_ptr(), wq.data_ptr(), _p(bias),
                                                              _p(residual), y.data_ptr(), n, h, w, cin, cout, ksize,
                                                              int(ups), float(pool_scale),
                                                              zero_page(x.device).data_ptr(), _stream()), nbytes)
    _native.check(st, 'conv2d_fprop_pooled')
    return y


FUSE_GN_STATS = _native.switch('VQK_FUSE_GN_STATS', '1') != '0'
if _native.switch('VQK_DETERMINISTIC', '') == '1':          # same as set_deterministic(True), from the environment
    DETERMINISTIC = True


X3_GNSTATS = _native.switch('VQK_X3_GNSTATS', '1') != '0'


def raw_conv_fprop_gnstats(x, wq, bias, residual, ups: bool, cout: int, groups: int, pool: bool = False,
                           pool_scale: float = 0.25, wlayout: int = 1):
    """3x3 conv (+ fused 2x2 pooling) whose drain also leaves the GroupNorm sums of its OUTPUT in the stream's GroupNorm
    workspace (vqk_conv2d_fprop_gnstats).  Returns y, or None when the problem is not served by the fused kernel
    (nothing launched): the caller runs the plain conv, and the next ``raw_gn_forward`` its own statistics pass."""
    _require_gpu(x)
    n, cin, h, w = x.shape
    s = 2 if ups else 1
    ho, wo = (h * s // 2, w * s // 2) if pool else (h * s, w * s)
    x3 = wlayout == 5 and x.dtype == torch.float32
    if x3 and (pool or DETERMINISTIC or not X3_GNSTATS or cout % 128 or (cout // groups) not in (4, 8, 16)):
        return None
    if not FUSE_GN_STATS or (x.dtype != torch.bfloat16 and not x3) or ho * wo <= 1024:
        return None
    if _HANDOFF.gn is not None:                                  # sums nobody claimed (the consumer was not a GroupNorm)
        _claim_presummed(x, -1)
    y = empty_nhwc(n, cout, ho, wo, x.dtype, x.device)
    ws = _gn_sum_target(x.device, n, groups, h * s * w * s)
    flops = 2.0 * n * h * s * w * s * cout * cin * 9
    if x3:
        # split-product mode: fp32 sums of the fp32 output from the accumulators (csrc/conv_x3.hip)
        nbytes = x.numel() * 4 + y.numel() * 4 * (2 if residual is not None else 1) + cout * cin * 9 * 4
        st = _timed(_fprop_kernel_name(x.dtype, 5), flops,
                    lambda: _native.lib().vqk_conv2d_fprop_x3_gnstats(x.data_ptr(), wq.data_ptr(), _p(bias), _p(residual), y.data_ptr(), n, h, w,
                                                                      cin, cout, int(ups), ws.data_ptr(), groups,
                                                                      zero_page(x.device).data_ptr(), _stream()), nbytes, exec_flops=3.0 * flops)
        if st == _native.ERR_SHAPE:
            return None
        _native.check(st, 'conv2d_fprop_x3_gnstats')
        return y
    nbytes = (x.numel() * x.element_size() + y.numel() * y.element_size()
              + (residual.numel() * residual.element_size() if residual is not None else 0) + cout * cin * 9 * x.element_size())
    st = _timed(_fprop_kernel_name(x.dtype, 1, (n, h * s, w * s, cin, cout, 0, x.dtype)), flops,
                lambda: _native.lib().vqk_conv2d_fprop_gnstats(dcode(x.dtype), x.data_ptr(), wq.data_ptr(), _p(bias),
                                                               _p(residual), y.data_ptr(), n, h, w, cin, cout, 3, int(ups),
                                                               int(pool), float(pool_scale), ws.data_ptr(), groups,
                                                               zero_page(x.device).data_ptr(), _stream()), nbytes)
    if st == _native.ERR_SHAPE:
        return None
    _native.check(st, 'conv2d_fprop_gnstats')
    return y


THIN_IN_GNSTATS = _native.switch('VQK_THIN_IN_GNSTATS', '1') != '0'


def raw_conv_thin_in_gnstats(x, wq, bias, cout: int, groups: int):
    """the 3x3 conv on the padded 3-channel image (the encoder's first conv) with the GroupNorm sums of its output in the
    stream's workspace (vqk_conv2d_thin_in_gnstats); None when not served (nothing launched)"""
    _require_gpu(x)
    n, cin, h, w = x.shape
    if (not FUSE_GN_STATS or not THIN_IN_GNSTATS or DETERMINISTIC or x.dtype != torch.bfloat16 or cin != 8 or h * w <= 1024):
        return None
    if _HANDOFF.gn is not None:
        _claim_presummed(x, -1)
    y = empty_nhwc(n, cout, h, w, x.dtype, x.device)
    ws = _gn_sum_target(x.device, n, groups, h * w)
    nbytes = x.numel() * x.element_size() + y.numel() * y.element_size()
    st = _timed('conv3x3_thin_in_kernel<bf16> (HBM)' + (f' {cin}->{cout}@{h}x{w} +gn-sums' if _EVENT_SHAPES else ''),
                0.0,
                lambda: _native.lib().vqk_conv2d_thin_in_gnstats(dcode(x.dtype), x.data_ptr(), wq.data_ptr(), _p(bias), y.data_ptr(),
                                                                 n, h, w, cout, ws.data_ptr(), groups, _stream()), nbytes)
    if st == _native.ERR_SHAPE:
        return None
    _native.check(st, 'conv2d_thin_in_gnstats')
    return y


UPS_PHASE = int(_native.switch('VQK_UPS_PHASE', '1'))      # 0 off, 1 forward + data gradient, 2 forward only
UPS_PHASE_WGRAD = _native.switch('VQK_UPS_PHASE_WGRAD', '1') != '0'    # the upsample convs' WEIGHT gradient in phase form too (round 5)


X3_PHASE = _native.switch('VQK_X3_PHASE', '1') != '0'         # the phase forms in the split-product mode too (conv_x3.hip NTAP = 4)


def phase_layout(dtype, x3: bool) -> int:
    """operand layout of the phase forms: 2 (bf16), 6 (fp32 tensors in the split-product mode), 0: no phase form in this mode"""
    if dtype == torch.bfloat16:
        return 2
    return 6 if (dtype == torch.float32 and x3 and X3_PHASE) else 0


def _phase_gn_ok(x, cout: int, gn_groups: int) -> bool:
    """the fused GroupNorm sums of a phase launch (fp32: conv_x3.hip's atomic sums, groups of 4 / 8 / 16 channels)"""
    if x.dtype == torch.bfloat16:
        return True
    return X3_GNSTATS and not DETERMINISTIC and (cout // gn_groups) in (4, 8, 16)


def _phase_name(dtype) -> str:
    return 'conv3x3_mx_kernel<bf16>' if dtype == torch.bfloat16 else _fprop_kernel_name(dtype, 5)


def raw_conv_ups_phase(x, wq4, bias, cout: int, backward: bool, gn_groups: int = 0):
    """The nearest-x2 upsample + 3x3 conv in phase form (vqk_conv2d_ups_phase: four 2x2-tap launches, 4/9 of the
    multiply-adds).  forward: x [N,Cin,h,w] -> [N,cout,2h,2w] (+ bias; gn_groups: also the GroupNorm sums of the result);
    backward: x = dy [N,C,2h,2w] -> dx [N,cout,h,w].  wq4: ``packed_weight(..., layout=phase_layout(...))`` (transpose for backward).
    Returns None when the kernel does not serve the problem (nothing launched)."""
    _require_gpu(x)
    if not UPS_PHASE or x.dtype not in (torch.bfloat16, torch.float32):      # (fp32 tensors: the split-product mode, layout 6)
        return None
    n, cin, hx, wx = x.shape
    h, w = (hx // 2, wx // 2) if backward else (hx, wx)
    if cout % 128 or cin % (64 if x.dtype == torch.bfloat16 else 32) or (backward and (hx % 2 or wx % 2)):
        return None
    y = empty_nhwc(n, cout, h if backward else 2 * h, w if backward else 2 * w, x.dtype, x.device)
    ws = None
    if gn_groups and not backward and FUSE_GN_STATS and 4 * h * w > 1024 and _phase_gn_ok(x, cout, gn_groups):
        if _HANDOFF.gn is not None:
            _claim_presummed(x, -1)
        ws = _gn_sum_target(x.device, n, gn_groups, 4 * h * w)
    flops = 2.0 * n * 4 * h * w * cout * cin * 9                 # ALGORITHMIC: the 3x3 conv over the upsampled image
    nbytes = x.numel() * x.element_size() + y.numel() * y.element_size() + cout * cin * 9 * x.element_size()
    f32 = x.dtype == torch.float32
    st = _timed(_phase_name(x.dtype) + (f' {cin}->{cout}@{y.shape[2]}x{y.shape[3]} phase-{"dgrad" if backward else "fwd"}' if _EVENT_SHAPES else ''), flops,
                lambda: _native.lib().vqk_conv2d_ups_phase(dcode(x.dtype), x.data_ptr(), wq4.data_ptr(), _p(bias), y.data_ptr(),
                                                           n, h, w, cin, cout, int(backward), _p(ws), gn_groups,
                                                           zero_page(x.device).data_ptr(), _stream()), nbytes,
                launches=1 if (_UPS_MERGE or f32) else 4, exec_flops=flops * (3.0 if f32 else 1.0) * 4.0 / 9.0)
    if st == _native.ERR_SHAPE:
        return None
    _native.check(st, 'conv2d_ups_phase')
    if ws is not None:
        _note_presummed(y, gn_groups)
    return y


POOLED_DGRAD_PHASE = _native.switch('VQK_POOLED_DGRAD_PHASE', '1') != '0'


def raw_conv_pooled_dgrad_phase(dy_pooled, weight, scale: float, x3: bool = False):
    """data gradient of a 3x3 conv followed by a 2x2 average pool from the POOLED gradient, in phase form
    (vqk_conv2d_pooled_dgrad_phase): dy_pooled [N, O, h, w] -> dx [N, I, 2h, 2w] = scale * (nearest-x2(dy_pooled) conv flip(W)^T),
    4/9 of the multiply-adds of the tap form.  None when the kernel does not serve the problem (nothing launched)."""
    _require_gpu(dy_pooled)
    lay = phase_layout(dy_pooled.dtype, x3)
    if not UPS_PHASE or not lay:
        return None
    o, i = weight.shape[0], weight.shape[1]
    n, c, h, w = dy_pooled.shape
    if c != o or i % 128 or o % (64 if lay == 2 else 32):
        return None
    dx = empty_nhwc(n, i, 2 * h, 2 * w, dy_pooled.dtype, dy_pooled.device)
    w4t = packed_weight(weight, i, o, dy_pooled.dtype, 3, True, lay)
    flops = 2.0 * n * 4 * h * w * o * i * 9                      # ALGORITHMIC: the 3x3 data gradient at full resolution
    es = dy_pooled.element_size()
    nbytes = dy_pooled.numel() * es + dx.numel() * es + o * i * 9 * es
    st = _timed(_phase_name(dy_pooled.dtype) + (f' {o}->{i}@{2 * h}x{2 * w} pooled-dgrad phase' if _EVENT_SHAPES else ''), flops,
                lambda: _native.lib().vqk_conv2d_pooled_dgrad_phase(dcode(dy_pooled.dtype), dy_pooled.data_ptr(), w4t.data_ptr(),
                                                                    dx.data_ptr(), n, h, w, o, i, float(scale),
                                                                    zero_page(dy_pooled.device).data_ptr(), _stream()), nbytes,
                exec_flops=flops * (3.0 if lay == 6 else 1.0) * 4.0 / 9.0)
    if st == _native.ERR_SHAPE:
        return None
    _native.check(st, 'conv2d_pooled_dgrad_phase')
    return dx


POOLED_FPROP_PHASE = _native.switch('VQK_POOLED_FPROP_PHASE', '1') != '0'
POOLED_WGRAD_PHASE = _native.switch('VQK_POOLED_WGRAD_PHASE', '0') != '0'      # measured +-0 in the step (27.18 / 27.23 against 27.21 / 27.26 ms: these launches sit under the GroupNorm backward): off
POOLED_FPROP_MIN_HW = int(_native.switch('VQK_POOLED_FPROP_MIN_HW', '4096'))      # full-resolution pixels per image from which it is used


def raw_conv_pooled_fprop_phase(x, weight, res_pooled, scale: float, gn_groups: int = 0, x3: bool = False):
    """scale * sumpool2x2(conv3x3(x, W)) + res_pooled as ONE 4x4 stride-2 launch (vqk_conv2d_pooled_fprop_phase): x [N, I, 2h, 2w]
    -> [N, O, h, w]; gn_groups: the GroupNorm sums of the result go to the stream's workspace.  None when not served."""
    _require_gpu(x)
    lay = phase_layout(x.dtype, x3)
    if not UPS_PHASE or not lay:
        return None
    o, i = weight.shape[0], weight.shape[1]
    n, c, hx, wx = x.shape
    if c != i or o % 128 or i % (64 if lay == 2 else 32) or hx % 2 or wx % 2:
        return None
    h, w = hx // 2, wx // 2
    y = empty_nhwc(n, o, h, w, x.dtype, x.device)
    ws = None
    if gn_groups and FUSE_GN_STATS and h * w > 1024 and _phase_gn_ok(x, o, gn_groups):
        if _HANDOFF.gn is not None:
            _claim_presummed(x, -1)
        ws = _gn_sum_target(x.device, n, gn_groups, h * w)
    w4 = packed_weight(weight, i, o, x.dtype, 3, False, lay)
    flops = 2.0 * n * hx * wx * o * i * 9                        # ALGORITHMIC: the 3x3 conv at full resolution
    es = x.element_size()
    nbytes = x.numel() * es + y.numel() * es * (2 if res_pooled is not None else 1) + o * i * 9 * es
    st = _timed(_phase_name(x.dtype) + (f' {i}->{o}@{hx}x{wx} pooled-fprop phase' if _EVENT_SHAPES else ''), flops,
                lambda: _native.lib().vqk_conv2d_pooled_fprop_phase(dcode(x.dtype), x.data_ptr(), w4.data_ptr(), _p(res_pooled), y.data_ptr(),
                                                                    n, h, w, i, o, float(scale), _p(ws), gn_groups,
                                                                    zero_page(x.device).data_ptr(), _stream()), nbytes,
                exec_flops=flops * (3.0 if lay == 6 else 1.0) * 4.0 / 9.0)
    if st == _native.ERR_SHAPE:
        return None
    _native.check(st, 'conv2d_pooled_fprop_phase')
    if ws is not None:
        _note_presummed(y, gn_groups, conv_hw=h * w)
    return y


# ---- gradient modes of ONE backward call (no process-wide flag) -------------------------------------------------------------
# `no_direct_grad` / `no_param_grads` describe a particular backward()/autograd.grad() call.  The custom Functions' backward methods
# run on the autograd engine's device thread, not on the caller's -- so the mode travels with the GRAPH TASK: the caller's
# contexts push onto a thread-local stack, `ops.backward` / `ops.autograd_grad` tag the graph task they start (a hook on the root
# runs first thing inside it and files the caller's mode under torch._C._current_graph_task_id()), and `_grad_modes()` inside a
# node's backward looks its own task up.  Two models stepping on two host threads -- two VQ-GAN steps included -- cannot see each
# other's modes (tests/test_gpu_two_models.py); rounds 1-5 kept these as module globals with save / restore.
_MODE_TLS = threading.local()
_TASK_MODES: dict = {}            # graph task id -> (direct_grad, param_grads)
_DEFAULT_MODES = (True, True)


def _mode_stack() -> list:
    st = getattr(_MODE_TLS, 'stack', None)
    if st is None:
        st = _MODE_TLS.stack = []
    return st


def _grad_modes() -> tuple:
    """(direct_grad, param_grads) of the backward call this code runs under"""
    tid = torch._C._current_graph_task_id()
    if tid != -1:
        m = _TASK_MODES.get(tid)
        if m is not None:
            return m
    st = getattr(_MODE_TLS, 'stack', None)          # (a backward executed on the calling thread itself; plain forward code)
    return st[-1] if st else _DEFAULT_MODES


class _mode_ctx:
    _index, _value = 0, False

    def __enter__(self):
        st = _mode_stack()
        cur = list(st[-1] if st else _DEFAULT_MODES)
        cur[self._index] = self._value
        st.append(tuple(cur))

    def __exit__(self, *exc):
        _mode_stack().pop()


class no_direct_grad(_mode_ctx):
    """context around ``ops.autograd_grad`` / ``ops.backward``: gradients are RETURNED through autograd instead of being accumulated
    into the flat arena (the adaptive generator weight asks for d loss / d last_layer, loss.py:80-96)"""
    _index, _value = 0, False


class no_param_grads(_mode_ctx):
    """context around ``ops.backward`` / ``ops.autograd_grad``: ConvActFn's backward computes the DATA gradient only.  The generator
    loss backpropagates through the discriminator for the sake of the decoder alone (the reference zeroes the discriminator
    gradients of that pass before its discriminator step, model.py:258); with the fake pass shared between the two halves of the
    step the discriminator's parameters must keep requires_grad, so freezing them is no option."""
    _index, _value = 1, False


def _tagged(roots, run):
    """run() -- a backward()/autograd.grad() over ``roots`` -- with the calling thread's gradient modes attached to its graph task"""
    st = getattr(_MODE_TLS, 'stack', None)
    mode = st[-1] if st else _DEFAULT_MODES
    if mode == _DEFAULT_MODES:
        return run()
    seen, hooks = [], []

    def tag(grad):
        tid = torch._C._current_graph_task_id()
        if tid != -1 and tid not in seen:
            _TASK_MODES[tid] = mode
            seen.append(tid)
        return None
    for r in roots:
        if torch.is_tensor(r) and r.requires_grad:
            hooks.append(r.register_hook(tag))
    try:
        return run()
    finally:
        for h in hooks:
            h.remove()
        for tid in seen:
            _TASK_MODES.pop(tid, None)


def backward(tensor, *args, **kwargs):
    """``tensor.backward(...)`` under the calling thread's ``no_direct_grad`` / ``no_param_grads`` contexts"""
    return _tagged((tensor,), lambda: tensor.backward(*args, **kwargs))


def autograd_grad(outputs, inputs, *args, **kwargs):
    """``torch.autograd.grad(...)`` under the calling thread's gradient-mode contexts"""
    roots = outputs if isinstance(outputs, (tuple, list)) else (outputs,)
    return _tagged(roots, lambda: torch.autograd.grad(outputs, inputs, *args, **kwargs))


def direct_grad(param):
    """The flat-arena gradient view of ``param`` when the HIP kernels may accumulate straight into it
    (FlatAdamW marks its parameters; the arena is zeroed once per step by ``zero_grad``), else None."""
    if _grad_modes()[0] and getattr(param, '_vqk_direct_grad', False) and param.grad is not None:
        return param.grad
    return None


_EDGE_WGRAD = _native.switch('VQK_EDGE_WGRAD', '1') != '0'
_EDGE_WS: dict = {}


def _edge_ws(device) -> torch.Tensor:
    """split-K workspace of vqk_conv2d_wgrad_edge, one per (device, stream): plain stores + ordered reduce, no atomics"""
    _stream()
    key = _wkey(device)
    ws = _EDGE_WS.get(key)
    if ws is None:
        ws = _EDGE_WS[key] = torch.empty(_native.lib().vqk_conv2d_wgrad_edge_ws_bytes() // 4, dtype=torch.float32, device=device)
    return ws


def edge_wgrad_served(x, dy, ksize: int, ups: bool) -> bool:
    """the K = 72 weight-gradient kernel of the two edge convs (padded 3-channel image / reconstruction) serves this problem"""
    n, cin, h, w = x.shape
    return (_EDGE_WGRAD and x.dtype == torch.bfloat16 and ksize == 3 and not ups and (cin, dy.shape[1]) in ((8, 128), (128, 8))
            and (h * w) % 128 == 0 and (w % 128 == 0 or 128 % w == 0))


def raw_split_pair(t) -> torch.Tensor:
    """fp32 nhwc [N,C,H,W] -> bf16 nhwc [N,2C,H,W] = (hi | lo) per pixel (vqk_split_pair_f32): hi = bf16(v), lo = bf16(v - hi)"""
    n, c, h, w = t.shape
    out = empty_nhwc(n, 2 * c, h, w, torch.bfloat16, t.device)
    st = _timed('split_pair_kernel (HBM)', 0.0,
                lambda: _native.lib().vqk_split_pair_f32(t.data_ptr(), out.data_ptr(), n * h * w, c, _stream()), t.numel() * 8.0)
    _native.check(st, 'split_pair')
    return out


_ABL_SKIP_SMALL_WGRAD = int(_native.switch('VQK_ABL_SKIP_SMALL_WGRAD', '0'))      # TIMING-ONLY ablation (tools/ab_env_multi.sh): 3x3 weight gradients on maps of <= this many pixels are not launched -- is their time hidden under the GroupNorm backward?


def raw_conv_wgrad(x, dy, ksize: int, ups: bool, out=None, thin_true: int = 8, x3: bool = False) -> torch.Tensor:
    """dw as fp32 with memory [Cout][k][k][Cin] (logical [Cout,Cin,k,k] channels_last); ``out``: accumulate
    into this (pre-existing) buffer instead of a fresh zeroed one.  ``thin_true`` < 8 (edge convs only, with ``out``): ``out`` is
    the parameter's own UNPADDED gradient (3 true channels on the thin side: vqk_conv2d_wgrad_edge_true)."""
    n, cin, h, w = x.shape
    cout = dy.shape[1]
    dw = out if out is not None else \
        torch.zeros((cout, ksize, ksize, cin), dtype=torch.float32, device=x.device).permute(0, 3, 1, 2)
    flops = 2.0 * n * dy.shape[2] * dy.shape[3] * cout * cin * ksize * ksize
    if _ABL_SKIP_SMALL_WGRAD and ksize == 3 and out is not None and dy.shape[2] * dy.shape[3] <= _ABL_SKIP_SMALL_WGRAD:
        return dw                                                # (wrong gradients: a timing experiment, never a training run)
    if edge_wgrad_served(x, dy, ksize, ups):
        # the two edge convs (padded 3-channel image / reconstruction): K = 72 GEMM, HBM-bound, workspace split-K
        ws = _edge_ws(x.device)
        nbytes = (x.numel() + dy.numel()) * 2
        st = _timed('conv3x3_wgrad_thin_kernel<bf16>' + (f' {cin}->{cout}@{h}x{w}' if _EVENT_SHAPES else ''), 0.0,
                    lambda: _native.lib().vqk_conv2d_wgrad_edge_true(dcode(x.dtype), x.data_ptr(), dy.data_ptr(), dw.data_ptr(),
                                                                     ws.data_ptr(), ws.numel() * 4, n, h, w, cin, cout, int(thin_true),
                                                                     zero_page(x.device).data_ptr(), _stream()), nbytes)
        _native.check(st, 'conv2d_wgrad_edge')
        return dw
    if thin_true != 8:
        raise RuntimeError('vqk: an unpadded weight-gradient target needs the edge-conv kernel')
    if (x3 and not X3_WGRAD_FOLD and x.dtype == torch.float32 and dy.dtype == torch.float32 and ksize == 3 and cin % 64 == 0
            and cout % 64 == 0 and dy.shape[3] % 8 == 0 and dy.shape[2] % 8 == 0 and not DETERMINISTIC):
        # split-product mode: both fp32 operands split in registers on their way into LDS, three products per staged fragment pair
        # (csrc/conv_x3.hip: conv3x3_wgrad_x3_kernel)
        phase = bool(ups) and X3_WGRAD_PHASE and h % 8 == 0 and w % 8 == 0      # (the library's own rule: include/vqk.h)
        st = _timed('conv3x3_wgrad_x3_kernel<f32 as 3 x bf16>' + (f' {cin}->{cout}@{dy.shape[2]}x{dy.shape[3]} k3{" phase" if phase else ""}' if _EVENT_SHAPES else ''), flops,
                    lambda: _native.lib().vqk_conv2d_wgrad_x3_f32(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), n, h, w, cin, cout, int(ups),
                                                                  1.0, _stream()), exec_flops=3.0 * flops * (4.0 / 9.0 if phase else 1.0))
        if st != _native.ERR_SHAPE:
            _native.check(st, 'conv2d_wgrad_x3_f32')
            return dw
    if (x3 and x.dtype == torch.float32 and dy.dtype == torch.float32 and ksize == 3 and cin % 64 == 0 and cout % 64 == 0
            and dy.shape[3] % 16 == 0 and dy.shape[2] % 8 == 0 and not DETERMINISTIC and _WGMX_ON):
        # split-product mode: both operands as (hi | lo) bf16 pair tensors, three tile classes of one launch of the bf16
        # matrix/auxiliary-wave weight-gradient kernel folded onto dW (csrc/conv_wgmx.hip, ConvGeom::fold)
        xp, dyp = raw_split_pair(x), raw_split_pair(dy)
        st = _timed('conv3x3_wgrad_mx_kernel<f32 as 3 x bf16>' + (f' {cin}->{cout}@{dy.shape[2]}x{dy.shape[3]} k3' if _EVENT_SHAPES else ''), flops,
                    lambda: _native.lib().vqk_conv2d_wgrad_x3(xp.data_ptr(), dyp.data_ptr(), dw.data_ptr(), n, h, w, cin, cout, int(ups),
                                                              1.0, zero_page(x.device).data_ptr(), _stream()), exec_flops=3.0 * flops)
        if st != _native.ERR_SHAPE:
            _native.check(st, 'conv2d_wgrad_x3')
            return dw
    if (ups and UPS_PHASE_WGRAD and _WGMX_ON and x.dtype == torch.bfloat16 and ksize == 3 and cin % 64 == 0 and cout % 64 == 0
            and w % 16 == 0 and h % 8 == 0 and not DETERMINISTIC):
        # the upsample conv's weight gradient in phase form: four 2x2-window launches on the LOW-resolution input, 4/9 of the
        # multiply-adds (csrc/conv_wgmx.hip, NT = 4)
        st = _timed('conv3x3_wgrad_mx_kernel<bf16>' + (f' {cin}->{cout}@{dy.shape[2]}x{dy.shape[3]} k{ksize} phase' if _EVENT_SHAPES else ''), flops,
                    lambda: _native.lib().vqk_conv2d_wgrad_ups_phase(dcode(x.dtype), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), n, h, w,
                                                                     cin, cout, 1.0, zero_page(x.device).data_ptr(), _stream()),
                    launches=4, exec_flops=flops * 4.0 / 9.0)
        if st != _native.ERR_SHAPE:
            _native.check(st, 'conv2d_wgrad_ups_phase')
            return dw
    mxw = (_WGMX_ON and x.dtype == torch.bfloat16 and ksize == 3 and cin % 64 == 0 and cout % 64 == 0
           and dy.shape[3] % 16 == 0 and dy.shape[2] % 8 == 0)
    kname = 'conv3x3_wgrad_mx_kernel<bf16>' if mxw else f'conv_wgrad_kernel<{"f32" if x.dtype == torch.float32 else "bf16"}>'
    if _EVENT_SHAPES:
        kname += f' {cin}->{cout}@{dy.shape[2]}x{dy.shape[3]} k{ksize}'
    st = _timed(kname, flops,
                lambda: _native.lib().vqk_conv2d_wgrad(dcode(x.dtype), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), n, h,
                                                       w, cin, cout, ksize, int(ups),
                                                       zero_page(x.device).data_ptr(), _stream()))
    _native.check(st, 'conv2d_wgrad')
    return dw


X3_WGRAD_PHASE = _native.switch('VQK_X3_WGRAD_PHASE', '1') != '0'     # (also a tuning slot of the library: the upsample conv's weight gradient in phase form)


def raw_conv_wgrad_pooled_x3(x, dy_pooled, scale: float, out) -> bool:
    """split-product mode: out += the weight gradient of a 3x3 conv FOLLOWED by a 2x2 average pool, from the pooled gradient, in phase
    form (vqk_conv2d_wgrad_x3_f32 with ups = 2: the 16 taps of the 4x4 stride-2 window at pooled resolution, 4/9 of the MFMAs).
    False: not served, nothing launched."""
    n, cin, h, w = x.shape
    cout = dy_pooled.shape[1]
    if (not X3_WGRAD_PHASE or x.dtype != torch.float32 or dy_pooled.dtype != torch.float32 or cin % 64 or cout % 64 or h % 16 or w % 16
            or DETERMINISTIC or tuple(dy_pooled.shape[2:]) != (h // 2, w // 2)):
        return False
    flops = 2.0 * n * h * w * cout * cin * 9
    st = _timed('conv3x3_wgrad_x3_kernel<f32 as 3 x bf16>' + (f' {cin}->{cout}@{h}x{w} k3 pooled-dy phase' if _EVENT_SHAPES else ''), flops,
                lambda: _native.lib().vqk_conv2d_wgrad_x3_f32(x.data_ptr(), dy_pooled.data_ptr(), out.data_ptr(), n, h, w, cin, cout, 2,
                                                              float(scale), _stream()), exec_flops=3.0 * flops * 4.0 / 9.0)
    if st == _native.ERR_SHAPE:
        return False
    _native.check(st, 'conv2d_wgrad_x3_f32')
    return True


def raw_conv_wgrad_pooled_dy(x, dy_pooled, scale: float, out) -> bool:
    """dW += scale * wgrad(x, unpool(dy_pooled)) without the unpooled tensor (vqk_conv2d_wgrad_pooled_dy); False when the
    kernel does not serve the problem (nothing launched)."""
    n, cin, h, w = x.shape
    cout = dy_pooled.shape[1]
    if not (_WGMX_ON and x.dtype == torch.bfloat16 and cin % 64 == 0 and cout % 64 == 0 and w % 16 == 0 and h % 8 == 0):
        return False
    flops = 2.0 * n * h * w * cout * cin * 9
    if POOLED_WGRAD_PHASE and not DETERMINISTIC and (h // 2) % 8 == 0 and (w // 2) % 16 == 0:
        # phase form (operands' roles swapped, csrc/conv_wgmx.hip): 4/9 of the multiply-adds
        st = _timed('conv3x3_wgrad_mx_kernel<bf16>' + (f' {cin}->{cout}@{h}x{w} pooled-dy phase' if _EVENT_SHAPES else ''), flops,
                    lambda: _native.lib().vqk_conv2d_wgrad_pooled_dy_phase(dcode(x.dtype), x.data_ptr(), dy_pooled.data_ptr(), out.data_ptr(),
                                                                           n, h // 2, w // 2, cin, cout, float(scale),
                                                                           zero_page(x.device).data_ptr(), _stream()),
                    exec_flops=flops * 4.0 / 9.0)
        if st != _native.ERR_SHAPE:
            _native.check(st, 'conv2d_wgrad_pooled_dy_phase')
            return True
    st = _timed('conv3x3_wgrad_mx_kernel<bf16>' + (f' {cin}->{cout}@{h}x{w} pooled-dy' if _EVENT_SHAPES else ''), flops,
                lambda: _native.lib().vqk_conv2d_wgrad_pooled_dy(dcode(x.dtype), x.data_ptr(), dy_pooled.data_ptr(), out.data_ptr(),
                                                                 n, h, w, cin, cout, float(scale),
                                                                 zero_page(x.device).data_ptr(), _stream()))
    if st == _native.ERR_SHAPE:
        return False
    _native.check(st, 'conv2d_wgrad_pooled_dy')
    return True


def raw_colsum(x2d_rows: int, c: int, x, out=None, lead: int | None = None, scale: float = 1.0) -> torch.Tensor:
    """out += scale * column sums; ``lead``: only the first ``lead`` columns are written (``out`` then has room for exactly those:
    the unpadded bias gradient)"""
    out = out if out is not None else torch.zeros(c, dtype=torch.float32, device=x.device)
    _native.check(_native.lib().vqk_colsum_lead(dcode(x.dtype), x.data_ptr(), x2d_rows, c, c if lead is None else lead, float(scale),
                                                out.data_ptr(), _stream()), 'colsum')
    return out


def raw_cast(src_f32, dtype) -> torch.Tensor:
    if dtype == torch.float32:
        return src_f32
    dst = torch.empty(src_f32.numel(), dtype=dtype, device=src_f32.device)
    _native.check(_native.lib().vqk_cast(src_f32.data_ptr(), dst.data_ptr(), dcode(dtype), src_f32.numel(), _stream()), 'cast')
    return dst


_GN_WS: dict = {}


GN_CLUSTER_MAX_HW = int(_native.switch('VQK_GN_CLUSTER_MAX_HW', '1024'))    # (mirrors the library's tuning slot: event bytes only)


def _gn_ws_doubles(n: int, c: int, groups: int) -> int:
    """sums [N][G][2] + one counter slot per sample + one ticket slot per (sample, 32-channel slice) (include/vqk.h:
    vqk_gn_backward_ws)"""
    return n * groups * 2 + n + n * (c // 32)


# The single-kernel CLUSTER form of the GroupNorm backward (csrc/norm.hip: gn_cluster_bwd_kernel) makes blocks wait for each other
# inside a launch.  Its progress argument (blocks dispatched in index order, a cluster = consecutive indices) holds for ONE such
# kernel at a time: two of them running concurrently -- two models stepped from two host threads, or two PROCESSES sharing a GPU --
# can each fill the slots the other's waiting blocks need; the bounded spin then gives up (vqk_gn_cluster_timeouts: wrong
# gradients, caught by check_kernel_health -- seen for real with two ranks on one GPU).  So the form belongs to ONE host thread's
# models per process: the first thread whose forward reaches a GroupNorm; every other thread's models take the two-kernel passes
# (the library takes the cluster form only when the workspace it is handed carries the ticket region).  Processes that share a
# GPU on purpose (bench.py's one-GPU dry run, the gloo two-rank test) set the tuning slot GN_CLUSTER_MAX_HW to 0.
_CLUSTER_OWNER = [None]
_CLUSTER_LOCK = threading.Lock()


def cluster_owner_ok() -> bool:
    """does the calling host thread own the cluster form of the GroupNorm backward?  (read at FORWARD time: the backward runs on the
    autograd engine's thread)"""
    me = threading.get_ident()
    own = _CLUSTER_OWNER[0]
    if own == me:
        return True
    with _CLUSTER_LOCK:
        own = _CLUSTER_OWNER[0]
        if own is None or not any(t.ident == own for t in threading.enumerate()):      # nobody yet / the owner thread is gone
            _CLUSTER_OWNER[0] = own = me
    return own == me


def _gn_red_size(red, n: int, c: int, groups: int, cluster_ok: bool) -> int:
    """the workspace size handed to the library: without the ticket region it takes the two-kernel passes (include/vqk.h)"""
    return red.numel() if cluster_ok else min(red.numel(), n * groups * 2 + n)


def _gn_cluster(dtype, hw: int, c: int, groups: int) -> bool:
    """does vqk_gn_backward_ws take its single-kernel cluster form for this map?  (bench statistics: 3 tensor passes, not 5)"""
    v = 4 if dtype == torch.float32 else 8
    sl = 64 if (c % 64 == 0 and 64 % (c // groups) == 0) else 32
    rows = 256 // (sl // v)
    small = 256 if dtype == torch.float32 else 512               # up to here: the one-block-per-slice kernels
    return (not DETERMINISTIC and small < hw <= GN_CLUSTER_MAX_HW and c % sl == 0 and sl % (c // groups) == 0
            and hw % (8 * rows) == 0)


_GN_PARTS: dict = {}


def _gn_parts(device, n_doubles: int) -> torch.Tensor:
    """deterministic mode: the per-tile slots a conv's drain leaves its GroupNorm sums in (+ N*G*2 doubles of scratch in front for
    their ordered total): plain stores, every slot written before it is read -- no zero protocol"""
    _stream()
    key = _wkey(device)
    ws = _GN_PARTS.get(key)
    if ws is None or ws.numel() < n_doubles:
        ws = _GN_PARTS[key] = torch.empty(max(n_doubles, 1 << 20), dtype=torch.float64, device=device)
    return ws


def _gn_sum_target(device, n: int, groups: int, conv_hw: int) -> torch.Tensor:
    """where the conv's drain puts the GroupNorm sums of its output: the stream's fp64 workspace (atomics, zero protocol), or in
    deterministic mode one slot per 256-pixel tile behind N*G*2 doubles of scratch (include/vqk.h: vqk_gn_forward_presummed_parts)"""
    if DETERMINISTIC:
        nblk = conv_hw // 256
        return _gn_parts(device, n * groups * 2 * (nblk + 1))[n * groups * 2:]
    return _gn_ws(device, n * groups * 2 + n)


def _gn_ws(device, n_doubles: int) -> torch.Tensor:
    """Persistent fp64 workspace of the GroupNorm kernels for the current stream (include/vqk.h: zero on entry, the
    consumer kernel leaves it zero again -> no memset launch per call)."""
    _stream()
    key = _wkey(device)
    ws = _GN_WS.get(key)
    if ws is None or ws.numel() < n_doubles:
        ws = torch.zeros(max(n_doubles, 8192), dtype=torch.float64, device=device)
        _GN_WS[key] = ws
    return ws


# A conv that was told "a GroupNorm with `groups` groups reads my output next" (``next_gn``) leaves the sums of its output
# in the stream's workspace (vqk_conv2d_fprop_gnstats) and notes the tensor here; the next ``raw_gn_forward`` on exactly
# that tensor claims them and skips its statistics pass.  Anything else arriving first finds the workspace dirty: it is
# cleared and the note dropped (the unfused sequence runs), so a changed call order costs a memset, never a wrong result.
#
# The hand-off is a HANDLE owned by the producing host thread (``_HANDOFF``, thread-local): (weak reference to the produced tensor,
# groups, workspace key, producer's conv resolution).  Producer and consumer run on one thread (a module's forward), the
# workspace the sums sit in is that thread's (``_wkey``): two models stepped from two Python threads -- or interleaved on one --
# cannot claim or clear each other's sums.
class _Handoff(threading.local):
    gn = None               # pending GroupNorm sums: (weakref(y), groups, workspace key, device, conv_hw)
    db = None               # pending bias column sum: (weakref(y), bias parameter)
    claimed_hw = 0          # conv resolution of the hand-off claimed last


_HANDOFF = _Handoff()


def pending_gn():
    """the calling thread's pending GroupNorm hand-off (None: nothing pending) -- tests and tooling"""
    return _HANDOFF.gn


def _note_presummed(y, groups: int, conv_hw: int = 0) -> None:
    """the note is keyed by the tensor OBJECT (weak reference), not by its address: a freed tensor whose memory the caching
    allocator hands to another tensor of the same shape can never claim stale sums"""
    _stream()
    _HANDOFF.gn = (weakref.ref(y), groups, _wkey(y.device), y.device, conv_hw if conv_hw else y.shape[2] * y.shape[3])


def _claim_presummed(x, groups: int) -> bool:
    p = _HANDOFF.gn
    if p is None:
        return False
    _HANDOFF.gn = None
    ref, g, wk, device, _hw = p
    stream = wk[1]
    _HANDOFF.claimed_hw = _hw
    if ref() is x and g == groups and stream == _stream():
        return True
    ws = None if DETERMINISTIC else _GN_WS.get(wk)                  # (deterministic mode: per-tile slots, nothing to clear)
    if ws is not None:                         # only the workspace the unclaimed sums were left in -- never another stream's
        if stream == _stream():
            ws.zero_()
        else:
            with torch.cuda.stream(torch.cuda.ExternalStream(stream, device=device)):
                ws.zero_()
    return False


def raw_gn_stats(x, groups: int, eps: float) -> torch.Tensor:
    n, c, h, w = x.shape
    acc = torch.zeros(n * groups * 2, dtype=torch.float64, device=x.device)
    stats = torch.empty(n * groups * 2, dtype=torch.float32, device=x.device)
    st = _native.lib().vqk_gn_stats(dcode(x.dtype), x.data_ptr(), n, h * w, c, groups, eps, acc.data_ptr(),
                                    stats.data_ptr(), _stream())
    _native.check(st, 'gn_stats')
    return stats


def raw_gn_apply(x, stats, w, b, groups: int, silu: bool) -> torch.Tensor:
    n, c, h, wd = x.shape
    y = torch.empty_like(x, memory_format=_CL)
    st = _native.lib().vqk_gn_apply(dcode(x.dtype), x.data_ptr(), stats.data_ptr(), w.data_ptr(), b.data_ptr(),
                                    y.data_ptr(), n, h * wd, c, groups, int(silu), _stream())
    _native.check(st, 'gn_apply')
    return y


def raw_gn_forward(x, w, b, groups: int, eps: float, silu: bool, presummed: bool = False, conv_hw: int = 0):
    """(y, stats): sums kernel + finalize-and-apply kernel on the persistent workspace; ``presummed``: the sums of x were
    left in the workspace by the conv that produced x (``raw_conv_fprop_gnstats``), only the apply pass runs"""
    n, c, h, wd = x.shape
    y = torch.empty_like(x, memory_format=_CL)
    stats = torch.empty(n * groups * 2, dtype=torch.float32, device=x.device)
    ws = _gn_ws(x.device, n * groups * 2 + n)
    nb = x.numel() * x.element_size()
    claimed = (not presummed) and _claim_presummed(x, groups)
    if claimed:
        conv_hw = _HANDOFF.claimed_hw                              # (a pooled producer: 4x the pixels of x)
    presummed = presummed or claimed
    if presummed and DETERMINISTIC:
        nblk = (conv_hw or h * wd) // 256
        buf = _gn_parts(x.device, n * groups * 2 * (nblk + 1))
        st = _timed('group_norm_fwd (HBM)' + (f' {c}@{h}x{wd} presummed' if _EVENT_SHAPES else ''), 0.0,
                    lambda: _native.lib().vqk_gn_forward_presummed_parts(dcode(x.dtype), x.data_ptr(), w.data_ptr(), b.data_ptr(),
                                                                         y.data_ptr(), stats.data_ptr(), buf[n * groups * 2:].data_ptr(),
                                                                         nblk, buf.data_ptr(), n, h * wd, c, groups, eps, int(silu),
                                                                         _stream()), 2 * nb)
        _native.check(st, 'gn_forward_presummed_parts')
        return y, stats
    if presummed:
        st = _timed('group_norm_fwd (HBM)' + (f' {c}@{h}x{wd} presummed' if _EVENT_SHAPES else ''), 0.0,
                    lambda: _native.lib().vqk_gn_forward_presummed(dcode(x.dtype), x.data_ptr(), w.data_ptr(), b.data_ptr(),
                                                                   y.data_ptr(), stats.data_ptr(), ws.data_ptr(), n, h * wd, c,
                                                                   groups, eps, int(silu), _stream()), 2 * nb)
        _native.check(st, 'gn_forward_presummed')
        return y, stats
    # algorithmic bytes: x read for the statistics, x read + y written by the apply pass (one read, one write on the
    # single-kernel path of the small maps)
    st = _timed('group_norm_fwd (HBM)' + (f' {c}@{h}x{wd}' if _EVENT_SHAPES else ''), 0.0,
                lambda: _native.lib().vqk_gn_forward(dcode(x.dtype), x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(),
                                                     stats.data_ptr(), ws.data_ptr(), n, h * wd, c, groups, eps, int(silu),
                                                     _stream()), (2 if h * wd <= 1024 else 3) * nb)
    _native.check(st, 'gn_forward')
    return y, stats


def raw_gn_backward(x, stats, w, b, dy, groups: int, silu: bool, dw=None, db=None, add=None, dx_colsum=None, out=None,
                    cluster_ok: bool | None = None):
    """``dx_colsum``: fp32 [C] buffer that also receives the per-channel sums of the dx written (the bias gradient of the conv
    that produced x: vqk_gn_backward_colsum); the caller checks :func:`gn_colsum_ok` first.  ``out``: dx goes here (batch slice)."""
    n, c, h, wd = x.shape
    dx = out if out is not None else torch.empty_like(x, memory_format=_CL)
    dw = dw if dw is not None else torch.zeros(c, dtype=torch.float32, device=x.device)
    db = db if db is not None else torch.zeros(c, dtype=torch.float32, device=x.device)
    _claim_presummed(x, -1)                                      # (clears a stale note + workspace; never matches)
    red = _gn_ws(x.device, _gn_ws_doubles(n, c, groups))
    nb = x.numel() * x.element_size()
    if cluster_ok is None:
        cluster_ok = cluster_owner_ok()                          # (direct callers: tests, tools)
    red_n = _gn_red_size(red, n, c, groups, cluster_ok)
    one_pass = h * wd <= 512 or (cluster_ok and _gn_cluster(x.dtype, h * wd, c, groups))
    passes = (3 if one_pass else 5) + (1 if add is not None else 0)     # x, dy (twice on the two-kernel path), dx, skip
    if dx_colsum is not None:
        st = _timed('group_norm_bwd (HBM)' + (f' {c}@{h}x{wd} +bias-grad' if _EVENT_SHAPES else ''), 0.0,
                    lambda: _native.lib().vqk_gn_backward_colsum(dcode(x.dtype), x.data_ptr(), stats.data_ptr(), w.data_ptr(),
                                                                 b.data_ptr(), dy.data_ptr(), dx.data_ptr(), dw.data_ptr(),
                                                                 db.data_ptr(), red.data_ptr(), red.numel(), n, h, wd, c, groups,
                                                                 int(silu), 0, _p(add), dx_colsum.data_ptr(), _stream()),
                    (5 + (1 if add is not None else 0)) * nb)
        _native.check(st, 'gn_backward_colsum')
        return dx, dw, db
    st = _timed('group_norm_bwd (HBM)' + (f' {c}@{h}x{wd}' if _EVENT_SHAPES else ''), 0.0,
                lambda: _native.lib().vqk_gn_backward_ws(dcode(x.dtype), x.data_ptr(), stats.data_ptr(), w.data_ptr(),
                                                         b.data_ptr(), dy.data_ptr(), dx.data_ptr(), dw.data_ptr(),
                                                         db.data_ptr(), red.data_ptr(), red_n, n, h, wd, c, groups,
                                                         int(silu), 0, _p(add), 0, 1.0, _stream()), passes * nb)
    _native.check(st, 'gn_backward')
    return dx, dw, db


# The bias gradient of a conv whose output feeds a ResBlock rides in that block's LAST GroupNorm-backward pass (the pass that
# writes d(block input) = the conv's dy): Conv2dFn.forward notes (output tensor, bias parameter), ResBlockFn.forward claims the
# note when that very tensor is its input, its backward hands the bias's arena gradient to the kernel and marks the bias as
# done; Conv2dFn.backward then skips its column-sum pass over dy (185 us for the 537-MB gradient of the last Upsample conv).
FUSE_BIAS_COLSUM = _native.switch('VQK_FUSE_BIAS_COLSUM', '1') != '0'
_DB_DONE: set = set()         # ids of bias parameters whose gradient a GroupNorm backward already summed (autograd thread; keyed
                              # by the parameter: two models never share an entry)


def gn_colsum_ok(h: int, w: int) -> bool:
    return FUSE_BIAS_COLSUM and not DETERMINISTIC and h * w > 1024


def _note_bias_colsum(y, bias) -> None:
    _HANDOFF.db = (weakref.ref(y), bias)


def _claim_bias_colsum(x):
    p, _HANDOFF.db = _HANDOFF.db, None
    if p is not None and p[0]() is x:
        return p[1]
    return None


def raw_gn_backward_pooled_add(x, stats, w, b, dy, groups: int, silu: bool, dw, db, add_pooled, add_scale: float,
                               cluster_ok: bool | None = None):
    """raw_gn_backward whose skip-branch addend is still at half resolution (vqk_gn_backward_pooled_add)"""
    n, c, h, wd = x.shape
    dx = torch.empty_like(x, memory_format=_CL)
    _claim_presummed(x, -1)
    red = _gn_ws(x.device, _gn_ws_doubles(n, c, groups))
    nb = x.numel() * x.element_size()
    if cluster_ok is None:
        cluster_ok = cluster_owner_ok()
    red_n = _gn_red_size(red, n, c, groups, cluster_ok)
    passes = 3.25 if (cluster_ok and _gn_cluster(x.dtype, h * wd, c, groups)) else 5.25
    st = _timed('group_norm_bwd (HBM)' + (f' {c}@{h}x{wd} pooled-add' if _EVENT_SHAPES else ''), 0.0,
                lambda: _native.lib().vqk_gn_backward_ws(dcode(x.dtype), x.data_ptr(), stats.data_ptr(), w.data_ptr(),
                                                         b.data_ptr(), dy.data_ptr(), dx.data_ptr(), dw.data_ptr(),
                                                         db.data_ptr(), red.data_ptr(), red_n, n, h, wd, c, groups,
                                                         int(silu), 1, 0, add_pooled.data_ptr(), float(add_scale), _stream()),
                passes * nb)
    _native.check(st, 'gn_backward_pooled_add')
    return dx


def raw_pool(x, scale: float) -> torch.Tensor:
    n, c, h, w = x.shape
    y = empty_nhwc(n, c, h // 2, w // 2, x.dtype, x.device)
    _native.check(_native.lib().vqk_pool2x2(dcode(x.dtype), x.data_ptr(), y.data_ptr(), n, h, w, c, scale, _stream()), 'pool2x2')
    return y


def raw_unpool(x, scale: float) -> torch.Tensor:
    n, c, h, w = x.shape
    y = empty_nhwc(n, c, h * 2, w * 2, x.dtype, x.device)
    _native.check(_native.lib().vqk_unpool2x2(dcode(x.dtype), x.data_ptr(), y.data_ptr(), n, h, w, c, scale, _stream()), 'unpool2x2')
    return y


def raw_preprocess(images, dtype, want_target: bool):
    """images [N,3,H,W] fp32 contiguous NCHW in [0,1] -> (x_pad [N,cpad,H,W] nhwc dtype, target fp32 or None)"""
    _require_gpu(images)
    n, c, h, w = images.shape
    if c != 3 or images.dtype != torch.float32:
        raise RuntimeError('vqk: preprocess expects fp32 images of shape [N,3,H,W]')
    images = images.contiguous()
    cp = epc(dtype)
    xp = empty_nhwc(n, cp, h, w, dtype, images.device)
    tgt = empty_nhwc(n, cp, h, w, torch.float32, images.device) if (want_target and dtype != torch.float32) else None
    st = _native.lib().vqk_preprocess(images.data_ptr(), xp.data_ptr(), dcode(dtype), _p(tgt), n, h, w, cp, _stream())
    _native.check(st, 'preprocess')
    return xp, (xp if (want_target and tgt is None) else tgt)


def random_crop_params(n: int, h: int, w: int, device, scale=(0.7, 1.0), generator=None):
    """Per-sample draws of RandomResizedCrop(scale, ratio=(1,1)) + RandomHorizontalFlip(p=0.5)
    (base_autoencoder.py:20-22), on the device: box [N,4] = (x0, y0, w, h) in pixels, flip [N] int32.
    Area fraction ~ U(scale), square boxes (ratio 1), top-left corner uniform over the valid range."""
    r = torch.rand(n, 4, device=device, generator=generator)
    area = scale[0] + (scale[1] - scale[0]) * r[:, 0]
    side = torch.sqrt(area)
    bw = torch.clamp(torch.floor(side * w + 0.5), 1, w)
    bh = torch.clamp(torch.floor(side * h + 0.5), 1, h)
    x0 = torch.floor(r[:, 1] * (w - bw + 1)).clamp(max=w - 1)
    y0 = torch.floor(r[:, 2] * (h - bh + 1)).clamp(max=h - 1)
    box = torch.stack([x0, y0, bw, bh], dim=1).to(torch.float32).contiguous()
    flip = (r[:, 3] < 0.5).to(torch.int32).contiguous()
    return box, flip


def raw_augment_preprocess(images, box, flip, dtype, want_target: bool):
    """raw_preprocess with the crop / flip augmentation fused in front (vqk_augment_preprocess)"""
    _require_gpu(images)
    n, c, h, w = images.shape
    if c != 3 or images.dtype != torch.float32:
        raise RuntimeError('vqk: preprocess expects fp32 images of shape [N,3,H,W]')
    images = images.contiguous()
    cp = epc(dtype)
    xp = empty_nhwc(n, cp, h, w, dtype, images.device)
    tgt = empty_nhwc(n, cp, h, w, torch.float32, images.device) if (want_target and dtype != torch.float32) else None
    st = _native.lib().vqk_augment_preprocess(images.data_ptr(), box.data_ptr(), flip.data_ptr(), xp.data_ptr(), dcode(dtype),
                                              _p(tgt), n, h, w, cp, _stream())
    _native.check(st, 'augment_preprocess')
    return xp, (xp if (want_target and tgt is None) else tgt)


# ------------------------------------------------------------------------------------------------------
# autograd functions
# ------------------------------------------------------------------------------------------------------
def _weight_mem(weight, cin_pad: int, cout_pad: int) -> torch.Tensor:
    """1-D fp32 view/copy of a logical [O,I,k,k] weight in [O][k][k][I] memory order, zero-padded to
    (cout_pad, cin_pad).  For a channels_last parameter with matching sizes this is a view."""
    o, i, k, _ = weight.shape
    w = weight.detach()
    if cin_pad == i and cout_pad == o:
        return w.permute(0, 2, 3, 1).reshape(-1)          # view for channels_last storage, copy otherwise
    wp = torch.zeros((cout_pad, k, k, cin_pad), dtype=torch.float32, device=w.device)
    wp[:o, :, :, :i] = w.permute(0, 2, 3, 1)
    return wp.reshape(-1)


class Conv2dFn(torch.autograd.Function):
    """y = act(conv(x (nearest-upsampled x2 if ups), W) + bias + residual).

    vqvae/modules/autoencoder.py:57-60,102-105,114,132,153,170.  Channel counts that are not a multiple
    of the 16-byte chunk (the 3-channel image / reconstruction) are zero-padded: x may carry more
    (zero) channels than the weight's I, and the output gets ``cout_pad`` channels, the extra ones
    identically zero."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, ups: bool, act: int, out_dtype, next_gn: int = 0):
        _require_gpu(x)
        x = nhwc(x)
        dt = x.dtype
        out_dtype = out_dtype or dt
        o, i, k, _ = weight.shape
        cin = x.shape[1]
        e = max(epc(dt), epc(out_dtype))
        cout_pad = -(-o // e) * e
        if cin < i or cin % epc(dt):
            raise RuntimeError(f'vqk: conv input has {cin} channels, weight expects {i}')
        n_img, _, h_in, w_in = x.shape
        layout = weight_layout(dt, n_img, h_in, w_in, cin, cout_pad, k, ups, out_dtype)
        wq = packed_weight(weight, cin, cout_pad, dt, k, False, layout)
        b32 = padded_vector(bias, cout_pad) if bias is not None else None
        res = nhwc(residual) if residual is not None else None
        y = None
        phase = False
        play = phase_layout(dt, X3)
        if (ups and k == 3 and act == 0 and res is None and out_dtype == dt and cout_pad == o and cin == i and UPS_PHASE
                and play and o % 128 == 0 and i % 128 == 0
                and weight_layout(dt, n_img, h_in, w_in, cin, cout_pad, 3, False) == (1 if play == 2 else 5)):
            # nearest x2 + 3x3 as four 2x2-tap convs on the low-resolution input (pre-summed weights)
            y = raw_conv_ups_phase(x, packed_weight(weight, cin, cout_pad, dt, 3, False, play), b32, cout_pad, False, next_gn)
            phase = y is not None
        if y is None and next_gn and k == 3 and act == 0 and layout in (1, 5) and out_dtype == dt and cout_pad % 128 == 0 and cout_pad == o:
            y = raw_conv_fprop_gnstats(x, wq, b32, res, ups, cout_pad, next_gn, wlayout=layout)
            if y is not None:
                _note_presummed(y, next_gn)
        if (y is None and next_gn and k == 3 and act == 0 and layout == 0 and cin == 8 and cout_pad == o and res is None and not ups
                and out_dtype == dt):
            y = raw_conv_thin_in_gnstats(x, wq, b32, cout_pad, next_gn)      # the encoder's first conv: sums for the first ResBlock's norm1
            if y is not None:
                _note_presummed(y, next_gn)
        if y is None:
            y = raw_conv_fprop(x, wq, b32, res, k, ups, act, out_dtype, cout_pad, layout)
        ctx.save_for_backward(x, weight, y if act == 1 else None)
        ctx.bias_ref, ctx.weight_ref = bias, weight
        ctx.cfg = (k, ups, act, o, i, cin, cout_pad, bias is not None, residual is not None, dt)
        ctx.phase = phase
        ctx.x3 = X3                                              # the backward multiplies the way the forward did
        if (bias is not None and act == 0 and cout_pad == o and out_dtype == dt and gn_colsum_ok(y.shape[2], y.shape[3])
                and direct_grad(bias) is not None):
            _note_bias_colsum(y, bias)                           # a ResBlock reading y next computes db in its backward
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, y = ctx.saved_tensors
        k, ups, act, o, i, cin, cout_pad, has_bias, has_res, dt = ctx.cfg
        dy = nhwc(dy)
        if act == 1:
            d2 = torch.empty_like(dy, memory_format=_CL)
            _native.check(_native.lib().vqk_tanh_backward(dcode(dy.dtype), dy.data_ptr(), y.data_ptr(), d2.data_ptr(),
                                                          dy.numel(), _stream()), 'tanh_backward')
            dy = d2
        dres = dy if has_res else None
        dyc = dy if dy.dtype == dt else nhwc(dy.to(dt))      # fp32 head output in bf16 mode: dtype cast only
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            n_img, _, h_out, w_out = dyc.shape
            layout = weight_layout(dt, n_img, h_out, w_out, cout_pad, cin, k, False, x3=ctx.x3)
            if ups and ctx.phase and UPS_PHASE == 1:             # data gradient in phase form: four 2x2-tap launches
                dx = raw_conv_ups_phase(dyc, packed_weight(weight, cin, cout_pad, dt, k, True, phase_layout(dt, ctx.x3)), None, cin, True)
            if dx is not None:
                pass
            elif ups and can_pool_epilogue(dt, cin, layout):
                wt = packed_weight(weight, cin, cout_pad, dt, k, True, layout)
                dx = raw_conv_fprop_pooled(dyc, wt, None, None, k, False, cin, 1.0)      # sum-pool in the epilogue
            else:
                wt = packed_weight(weight, cin, cout_pad, dt, k, True, layout)
                dx = raw_conv_fprop(dyc, wt, None, None, k, False, 0, dt, cin, layout)
                if ups:
                    dx = raw_pool(dx, 1.0)
        padded = cin != i or cout_pad != o
        if ctx.needs_input_grad[1]:
            tgt = None if padded else direct_grad(ctx.weight_ref)
            thin = 8
            if padded and edge_wgrad_served(x, dyc, k, ups) and (cin == i or cout_pad == o):
                # the 3-channel edge convs (image in, reconstruction out): the kernel drops the zero-padded channels itself and adds
                # the TRUE gradient to the arena -- no zero-filled padded temporary, no slice, no AccumulateGrad pass
                tgt = direct_grad(ctx.weight_ref)
                thin = (i if cin != i else o) if tgt is not None else 8
            if (tgt is not None and CONV_WGRAD_SIDE and OVERLAP_WGRAD and k == 3 and dt == torch.bfloat16 and thin == 8
                    and dyc.shape[2] * dyc.shape[3] >= CONV_WGRAD_SIDE_MIN_HW and not DETERMINISTIC):
                # a conv outside a ResBlock (the decoder's Upsample convs: 0.9 ms of weight gradients per step): the gradient goes
                # to the arena and only the optimizer reads it -- issued on the side stream behind the data gradient; the next
                # ResBlock's backward (or join_side_streams at the end of the backward) joins it
                main, side = torch.cuda.current_stream(), _side_stream(x.device)
                _side_after(side, _fork_point(main))
                with torch.cuda.stream(side):
                    raw_conv_wgrad(x, dyc, k, ups, out=tgt, thin_true=thin, x3=ctx.x3)
                _SIDE_PENDING.add(x.device)
                x.record_stream(side); dyc.record_stream(side)
            else:
                dw = raw_conv_wgrad(x, dyc, k, ups, out=tgt, thin_true=thin, x3=ctx.x3)
            if tgt is not None:
                dw = None                                        # already accumulated in the flat arena
            elif padded:
                dw = dw[:o, :i]
        if has_bias and ctx.needs_input_grad[2] and id(ctx.bias_ref) in _DB_DONE:
            _DB_DONE.discard(id(ctx.bias_ref))                   # the consumer's GroupNorm backward summed dy's columns already
        elif has_bias and ctx.needs_input_grad[2]:
            n, c, h, w = dyc.shape
            tgt = direct_grad(ctx.bias_ref) if (cout_pad == c) else None     # (padded output channels: only the true ones are written)
            db = raw_colsum(n * h * w, c, dyc, out=tgt, lead=o if tgt is not None else None)
            db = None if tgt is not None else db[:o]
        return dx, dw, db, dres, None, None, None, None


def conv2d(x, weight, bias=None, residual=None, ups: bool = False, act: int = 0, out_dtype=None, next_gn: int = 0):
    return Conv2dFn.apply(x, weight, bias, residual, ups, act, out_dtype, next_gn)


class GroupNormSiLUFn(torch.autograd.Function):
    """y = [silu](GroupNorm(x)) with the reference's unbiased variance (autoencoder.py:25-39)."""

    @staticmethod
    def forward(ctx, x, weight, bias, groups: int, eps: float, silu: bool):
        _require_gpu(x)
        x = nhwc(x)
        w = weight.detach().reshape(-1).contiguous()
        b = bias.detach().reshape(-1).contiguous()
        y, stats = raw_gn_forward(x, w, b, groups, eps, silu)
        ctx.save_for_backward(x, stats, w, b)
        ctx.params = (weight, bias)
        ctx.cfg = (groups, silu, weight.shape, bias.shape)
        ctx.cluster_ok = cluster_owner_ok()
        return y

    @staticmethod
    def backward(ctx, dy):
        x, stats, w, b = ctx.saved_tensors
        groups, silu, wshape, bshape = ctx.cfg
        tw, tb = direct_grad(ctx.params[0]), direct_grad(ctx.params[1])
        direct = tw is not None and tb is not None
        dx, dw, db = raw_gn_backward(x, stats, w, b, nhwc(dy), groups, silu, tw if direct else None,
                                     tb if direct else None, cluster_ok=ctx.cluster_ok)
        if direct:
            return dx, None, None, None, None, None
        return dx, dw.view(wshape), db.view(bshape), None, None, None


POOLED_BWD = _native.switch('VQK_POOLED_BWD', '1') != '0'      # ResBlock + fused avg-pool: the backward keeps the gradient pooled
OVERLAP_WGRAD = _native.switch('VQK_OVERLAP_WGRAD', '1') == '1'
SHORTCUT_WGRAD_SIDE = _native.switch('VQK_SHORTCUT_WGRAD_SIDE', '0') == '1'   # ResBlock shortcut: weight gradient on the side stream (measured +-0: 28.26 / 28.18 against 28.17 / 28.19 ms -- off)
OVERLAP_MODE = int(_native.switch('VQK_OVERLAP_MODE', '3'))
WGRAD_NOJOIN_HW = int(_native.switch('VQK_WGRAD_NOJOIN_HW', '0'))      # ResBlocks on maps of <= this many pixels: weight gradients joined at the END of the backward
OVERLAP_WAIT_MIN_HW = int(_native.switch('VQK_OVERLAP_WAIT_MIN_HW', '0'))   # maps below this many pixels: dgrad1 does not wait for wgrad2
OVERLAP_STREAM_BLOCKS = int(_native.switch('VQK_OVERLAP_STREAM_BLOCKS', '512'))
OVERLAP_WGRAD_BLOCKS = int(_native.switch('VQK_OVERLAP_WGRAD_BLOCKS', '320'))
OVERLAP_WGRAD_BLOCKS_HI = int(_native.switch('VQK_OVERLAP_WGRAD_BLOCKS_HI', '256'))   # the 256x256 levels: GroupNorm-bound in the backward -- the weight gradient on half the CUs (swept 192 / 224 / 256 / 288 / 320 / 448: -0.15 ms at 256)
OVERLAP_WGRAD_BLOCKS_LO = int(_native.switch('VQK_OVERLAP_WGRAD_BLOCKS_LO', str(OVERLAP_WGRAD_BLOCKS)))     # maps of <= 32x32: the GroupNorm backward beside the weight gradient is ONE short kernel
OVERLAP_WGRAD_BLOCKS_MID = int(_native.switch('VQK_OVERLAP_WGRAD_BLOCKS_MID', str(OVERLAP_WGRAD_BLOCKS)))   # the 128x128 levels   # swept 192...512 with the 8x16-patch wgrad: flat 224...320
_SIDE_STREAMS: dict = {}


def _side_stream(device) -> torch.cuda.Stream:
    st = _SIDE_STREAMS.get(device)
    if st is None:
        st = _SIDE_STREAMS[device] = torch.cuda.Stream(device=device)
    return st


def aux_stream(device, tag: str) -> torch.cuda.Stream:
    """a named extra stream per device (independent branches of a step, e.g. LPIPS next to the discriminator)"""
    st = _SIDE_STREAMS.get((device, tag))
    if st is None:
        st = _SIDE_STREAMS[(device, tag)] = torch.cuda.Stream(device=device)
    return st


# hipGraph replay maps the captured nodes to hardware queues by a depth-first walk in which the FIRST successor of a node
# (in capture order) inherits its queue and every further successor gets another one.  The weight-gradient launch is
# captured right behind the data-gradient conv it forks from, so the replay runs conv -> wgrad -> conv on one queue and the
# GroupNorm-backward chain on the other: every conv -> GroupNorm -> conv hop crosses queues (84 gaps of ~10 us,
# profiles/round3_step_timeline.txt).  CHAIN_FIRST = 1 captures the chain's next kernel first (the side stream then waits on
# an event recorded at the fork point): the replay does put the whole chain on one queue (profiles/round4_chain_first_ab.txt:
# 311 + 40 kernels instead of 275 + 76) -- and the step is SLOWER, 29.8-30.2 against 29.1-29.3 ms same box, with every cap /
# join variant tried: the kernel that reaches the chip first takes the CUs, the GroupNorm pass floods all 256 and the
# weight gradient (one 512-thread, 120-KiB block per CU) only gets in as its blocks retire (wgrad 11.3 instead of 9.9 ms,
# main queue waits 2.0 ms on it).  The weight gradient has to be launched first; the 10-us hops are the price.
CHAIN_FIRST = _native.switch('VQK_CHAIN_FIRST', '0') == '1'


CONV_WGRAD_SIDE = _native.switch('VQK_CONV_WGRAD_SIDE', '0') == '1'
CONV_WGRAD_SIDE_MIN_HW = int(_native.switch('VQK_CONV_WGRAD_SIDE_MIN_HW', '1024'))
_SIDE_PENDING: set = set()      # devices whose side stream carries work no ResBlock backward has joined yet


def join_side_streams() -> None:
    """the current stream waits for side-stream work that was issued outside a ResBlock's backward (Conv2dFn's weight gradient):
    called at the end of a backward, before the gradients are reduced / the optimizer steps"""
    for dev in list(_SIDE_PENDING):
        torch.cuda.current_stream(dev).wait_stream(_side_stream(dev))
    _SIDE_PENDING.clear()


def _fork_point(main):
    ev = torch.cuda.Event()
    ev.record(main)
    return ev


def _side_after(side, fork) -> None:
    side.wait_event(fork)


def _wgrad_cap(hw: int) -> int:
    """grid cap of the weight-gradient kernel next to the GroupNorm backward, per resolution level"""
    if hw <= 1024:
        return OVERLAP_WGRAD_BLOCKS_LO
    return OVERLAP_WGRAD_BLOCKS_HI if hw >= 65536 else OVERLAP_WGRAD_BLOCKS_MID if hw >= 16384 else OVERLAP_WGRAD_BLOCKS


class ResBlockFn(torch.autograd.Function):
    """One pre-activation residual block (autoencoder.py:63-77) as a single autograd node:
    GN+SiLU -> 3x3 -> GN+SiLU -> 3x3 (+ skip, optionally through a 1x1), with a hand-scheduled backward whose last
    GroupNorm-backward pass adds the skip-branch gradient in the same sweep (no separate add kernel, no autograd
    bookkeeping for eight intermediate nodes)."""

    @staticmethod
    def forward(ctx, x, n1w, n1b, c1w, n2w, n2b, c2w, scw, groups: int, eps: float, pool: bool = False, next_gn: int = 0):
        _require_gpu(x)
        ctx.db_param = _claim_bias_colsum(x)                     # x = output of a conv with a bias: its db rides in our backward
        x = nhwc(x)
        dt = x.dtype
        n, cin, h, w = x.shape
        cout = c1w.shape[0]
        if cin % epc(dt) or cout % epc(dt):
            raise RuntimeError('vqk: ResBlock channels must be whole 16-byte chunks')
        w1 = n1w.detach().reshape(-1).contiguous(); b1 = n1b.detach().reshape(-1).contiguous()
        w2 = n2w.detach().reshape(-1).contiguous(); b2 = n2b.detach().reshape(-1).contiguous()
        a1, st1 = raw_gn_forward(x, w1, b1, groups, eps, True)
        l1 = weight_layout(dt, n, h, w, cin, cout, 3, False)
        wq1 = packed_weight(c1w, cin, cout, dt, 3, False, l1)
        # the first conv's drain also sums its output for the second GroupNorm (no statistics pass over r1)
        r1 = raw_conv_fprop_gnstats(a1, wq1, None, None, False, cout, groups, wlayout=l1) if l1 in (1, 5) and cout % 128 == 0 else None
        fused = r1 is not None
        if not fused:
            r1 = raw_conv_fprop(a1, wq1, None, None, 3, False, 0, dt, cout, l1)
        a2, st2 = raw_gn_forward(r1, w2, b2, groups, eps, True, presummed=fused)
        skip = x
        if scw is not None:
            lsc = weight_layout(dt, n, h, w, cin, cout, 1, False)
            skip = raw_conv_fprop(x, packed_weight(scw, cin, cout, dt, 1, False, lsc), None, None, 1,
                                  False, 0, dt, cout, lsc)
        l2 = weight_layout(dt, n, h, w, cout, cout, 3, False)
        wq2 = packed_weight(c2w, cout, cout, dt, 3, False, l2)
        out = None
        if (pool and POOLED_FPROP_PHASE and cout % 128 == 0 and h * w >= POOLED_FPROP_MIN_HW
                and ((dt == torch.bfloat16 and l2 == 1) or (l2 == 5 and phase_layout(dt, X3) and h % 16 == 0 and w % 32 == 0))):
            # conv2 + the level's average pool as the 4x4 stride-2 conv it is (4/9 of the multiply-adds); the skip is pooled by its
            # own memory-bound pass (the launch adds a residual at its OUTPUT resolution)
            out = raw_conv_pooled_fprop_phase(a2, c2w, raw_pool(skip, 0.25), 0.25, next_gn, x3=X3)
        if out is not None:
            pass
        elif next_gn and l2 in (1, 5) and cout % 128 == 0:       # the sums for the GroupNorm that reads `out` next
            out = raw_conv_fprop_gnstats(a2, wq2, None, skip, False, cout, next_gn, pool=pool, pool_scale=0.25, wlayout=l2)
            if out is not None:
                _note_presummed(out, next_gn, conv_hw=h * w)
        if out is not None:
            pass
        elif pool and can_pool_epilogue(dt, cout, l2):
            out = raw_conv_fprop_pooled(a2, wq2, None, skip, 3, False, cout, 0.25)   # the level's avg-pool, fused
        else:
            out = raw_conv_fprop(a2, wq2, None, skip, 3, False, 0, dt, cout, l2)
            if pool:
                out = raw_pool(out, 0.25)
        ctx.save_for_backward(x, st1, a1, r1, st2, a2, w1, b1, w2, b2)
        ctx.params = (n1w, n1b, c1w, n2w, n2b, c2w, scw)
        ctx.cfg = (groups, cin, cout, pool)
        ctx.x3 = X3
        ctx.cluster_ok = cluster_owner_ok()
        return out

    @staticmethod
    def backward(ctx, dout):
        x, st1, a1, r1, st2, a2, w1, b1, w2, b2 = ctx.saved_tensors
        n1w, n1b, c1w, n2w, n2b, c2w, scw = ctx.params
        groups, cin, cout, pool = ctx.cfg
        x3 = ctx.x3
        dt = x.dtype
        dout = nhwc(dout)
        n, _, h, w = x.shape
        t1, t2 = direct_grad(c1w), direct_grad(c2w)
        tw1, tb1, tw2, tb2 = direct_grad(n1w), direct_grad(n1b), direct_grad(n2w), direct_grad(n2b)
        if (pool and POOLED_BWD and scw is None and dt == torch.bfloat16 and t1 is not None and t2 is not None
                and None not in (tw1, tb1, tw2, tb2) and OVERLAP_WGRAD and h * w > 1024 and cout % 128 == 0 and w % 16 == 0 and h % 8 == 0
                and weight_layout(dt, n, h // 2, w // 2, cout, cout, 3, True) == 1 and _MX_ON and _WGMX_ON):
            # The gradient of the fused avg-pool stays at HALF resolution: conv2's data gradient reads it through the nearest-x2
            # addressing of the halo DMA (the pool's 0.25 in the drain), conv2's weight gradient and norm1's skip addend read
            # the pooled pixel of each 2x2 block -- no unpool pass, a quarter of the gradient bytes for three consumers.
            main, side = torch.cuda.current_stream(), _side_stream(x.device)
            lib = _native.lib()
            lib.vqk_conv_set_block_caps(OVERLAP_STREAM_BLOCKS, _wgrad_cap(h * w))
            try:
                lay = weight_layout(dt, n, h // 2, w // 2, cout, cout, 3, True)
                wt2 = packed_weight(c2w, cout, cout, dt, 3, True, lay)
                d_a2 = raw_conv_pooled_dgrad_phase(dout, c2w, 0.25) if POOLED_DGRAD_PHASE else None
                if d_a2 is None:
                    d_a2 = _conv_general_raw(dout, wt2, None, None, cout, 3, 1, 1, 1, h, w, 0, 0.25, 1.0, dt, lay)
                fork = _fork_point(main)
                if not CHAIN_FIRST:
                    _side_after(side, fork)
                    with torch.cuda.stream(side):
                        if not raw_conv_wgrad_pooled_dy(a2, dout, 0.25, t2):
                            raise RuntimeError('vqk: pooled weight gradient not served for an eligible shape')
                d_r1, _, _ = raw_gn_backward(r1, st2, w2, b2, d_a2, groups, True, tw2, tb2, cluster_ok=ctx.cluster_ok)
                if CHAIN_FIRST:
                    _side_after(side, fork)
                    with torch.cuda.stream(side):
                        if not raw_conv_wgrad_pooled_dy(a2, dout, 0.25, t2):
                            raise RuntimeError('vqk: pooled weight gradient not served for an eligible shape')
                if OVERLAP_MODE == 3:
                    main.wait_stream(side)
                lay1 = weight_layout(dt, n, h, w, cout, cin, 3, False)
                d_a1 = raw_conv_fprop(d_r1, packed_weight(c1w, cin, cout, dt, 3, True, lay1), None, None, 3, False, 0, dt, cin, lay1)
                fork = _fork_point(main)
                if not CHAIN_FIRST:
                    _side_after(side, fork)
                    with torch.cuda.stream(side):
                        raw_conv_wgrad(a1, d_r1, 3, False, out=t1, x3=x3)
                dx = raw_gn_backward_pooled_add(x, st1, w1, b1, d_a1, groups, True, tw1, tb1, dout, 0.25, cluster_ok=ctx.cluster_ok)
                if CHAIN_FIRST:
                    _side_after(side, fork)
                    with torch.cuda.stream(side):
                        raw_conv_wgrad(a1, d_r1, 3, False, out=t1, x3=x3)
                main.wait_stream(side)
            finally:
                lib.vqk_conv_set_block_caps(0, 0)
            return dx, None, None, None, None, None, None, None, None, None, None, None
        d_a2_pre = dout_p = None
        if pool:
            if x3 and POOLED_DGRAD_PHASE and h % 16 == 0 and w % 32 == 0:
                # split-product mode: conv2's data gradient from the POOLED gradient in phase form (4/9 of the multiply-adds); so is its
                # weight gradient (wgrad_c2 below); the skip still reads the un-pooled gradient
                d_a2_pre = raw_conv_pooled_dgrad_phase(dout, c2w, 0.25, x3=True)
            dout_p = dout if x3 else None
            dout = raw_unpool(dout, 0.25)                   # backward of the fused avg-pool

        def wgrad_c2(tgt):
            if dout_p is not None and raw_conv_wgrad_pooled_x3(a2, dout_p, 0.25, tgt):
                return
            raw_conv_wgrad(a2, dout, 3, False, out=tgt, x3=x3)

        def conv_bwd(inp, dy, wparam, k, ci, co, need_dx=True, need_dw=True):
            dx = None
            if need_dx:
                lay = weight_layout(dt, n, h, w, co, ci, k, False, x3=x3)
                wt = packed_weight(wparam, ci, co, dt, k, True, lay)
                dx = raw_conv_fprop(dy, wt, None, None, k, False, 0, dt, ci, lay)
            if not need_dw:
                return dx, None
            tgt = direct_grad(wparam)
            dw = raw_conv_wgrad(inp, dy, k, False, out=tgt, x3=x3)
            return dx, (None if tgt is not None else dw)

        def gn_bwd(inp, st, wv, bv, dy, wparam, bparam, add=None, colsum_of=None):
            tw, tb = direct_grad(wparam), direct_grad(bparam)
            direct = tw is not None and tb is not None
            cs = None
            if colsum_of is not None and gn_colsum_ok(h, w) and not torch.is_grad_enabled():
                cs = direct_grad(colsum_of)
                if cs is not None:
                    _DB_DONE.add(id(colsum_of))
            dx, dw, db = raw_gn_backward(inp, st, wv, bv, dy, groups, True, tw if direct else None,
                                         tb if direct else None, add=add, dx_colsum=cs, cluster_ok=ctx.cluster_ok)
            if direct:
                return dx, None, None
            return dx, dw.view(wparam.shape), db.view(bparam.shape)

        if OVERLAP_WGRAD and t1 is not None and t2 is not None:
            # the two weight-gradient convs (MFMA-bound, results only needed by the optimizer) run on a side stream,
            # one block per CU, next to the data-gradient convs and the memory-bound GroupNorm backward passes
            main, side = torch.cuda.current_stream(), _side_stream(x.device)
            lib = _native.lib()
            lib.vqk_conv_set_block_caps(OVERLAP_STREAM_BLOCKS, _wgrad_cap(h * w))
            try:
                if OVERLAP_MODE == 1:
                    side.wait_stream(main)
                    with torch.cuda.stream(side):
                        wgrad_c2(t2)
                    d_a2 = d_a2_pre if d_a2_pre is not None else conv_bwd(a2, dout, c2w, 3, cout, cout, need_dw=False)[0]
                else:                                    # wgrad starts behind the dgrad: it overlaps GroupNorm only
                    d_a2 = d_a2_pre if d_a2_pre is not None else conv_bwd(a2, dout, c2w, 3, cout, cout, need_dw=False)[0]
                    fork = _fork_point(main)
                    if not CHAIN_FIRST:
                        _side_after(side, fork)
                        with torch.cuda.stream(side):
                            wgrad_c2(t2)
                d_r1, dn2w, dn2b = gn_bwd(r1, st2, w2, b2, d_a2, n2w, n2b)
                if OVERLAP_MODE != 1 and CHAIN_FIRST:
                    _side_after(side, fork)
                    with torch.cuda.stream(side):
                        wgrad_c2(t2)
                if OVERLAP_MODE == 1:
                    side.wait_stream(main)
                    with torch.cuda.stream(side):
                        raw_conv_wgrad(a1, d_r1, 3, False, out=t1, x3=x3)
                    d_a1, _ = conv_bwd(a1, d_r1, c1w, 3, cin, cout, need_dw=False)
                else:
                    if OVERLAP_MODE == 3 and h * w >= OVERLAP_WAIT_MIN_HW:
                        main.wait_stream(side)           # dgrad1 alone on the chip
                    d_a1, _ = conv_bwd(a1, d_r1, c1w, 3, cin, cout, need_dw=False)
                    fork = _fork_point(main)
                    if not CHAIN_FIRST:
                        _side_after(side, fork)
                        with torch.cuda.stream(side):
                            raw_conv_wgrad(a1, d_r1, 3, False, out=t1, x3=x3)
                dskip, dwsc = dout, None
                if scw is not None:
                    tsc = direct_grad(scw) if SHORTCUT_WGRAD_SIDE else None
                    if tsc is not None:
                        # the 1x1 shortcut: its data gradient (the skip addend of the GroupNorm pass below) on the main stream, its
                        # WEIGHT gradient -- needed by the optimizer only -- behind conv1's on the side stream, next to that pass
                        # (it ran on the main stream before: 0.42 ms per step on the critical path, profiles/round4_step_timeline.txt)
                        dskip, _ = conv_bwd(x, dout, scw, 1, cin, cout, need_dw=False)
                        fork_sc = _fork_point(main)
                        _side_after(side, fork_sc)
                        with torch.cuda.stream(side):
                            raw_conv_wgrad(x, dout, 1, False, out=tsc, x3=x3)
                    else:
                        dskip, dwsc = conv_bwd(x, dout, scw, 1, cin, cout)
                dx, dn1w, dn1b = gn_bwd(x, st1, w1, b1, d_a1, n1w, n1b, add=dskip, colsum_of=ctx.db_param)
                if OVERLAP_MODE != 1 and CHAIN_FIRST:
                    _side_after(side, fork)
                    with torch.cuda.stream(side):
                        raw_conv_wgrad(a1, d_r1, 3, False, out=t1, x3=x3)
                if h * w <= WGRAD_NOJOIN_HW:
                    # small maps: the block's GroupNorm backward is ONE short kernel, the weight gradients outlast it (measured: not
                    # launching them at all on <= 32^2 maps saves 1.23 ms of a 27.8-ms step, profiles/round6_small_wgrad_ab.txt) --
                    # they are joined at the end of the backward (join_side_streams) instead of at the end of the block and run
                    # beside the next blocks' kernels, which leave CUs free on these maps
                    _SIDE_PENDING.add(x.device)
                    for t in (a1, a2, dout, d_r1) + ((dout_p,) if dout_p is not None else ()):
                        t.record_stream(side)
                else:
                    main.wait_stream(side)
            finally:
                lib.vqk_conv_set_block_caps(0, 0)
            return dx, dn1w, dn1b, None, dn2w, dn2b, None, dwsc, None, None, None, None
        d_a2, dw2 = conv_bwd(a2, dout, c2w, 3, cout, cout, need_dx=d_a2_pre is None)
        if d_a2_pre is not None:
            d_a2 = d_a2_pre
        d_r1, dn2w, dn2b = gn_bwd(r1, st2, w2, b2, d_a2, n2w, n2b)
        d_a1, dw1 = conv_bwd(a1, d_r1, c1w, 3, cin, cout)
        dskip, dwsc = dout, None
        if scw is not None:
            dskip, dwsc = conv_bwd(x, dout, scw, 1, cin, cout)
        dx, dn1w, dn1b = gn_bwd(x, st1, w1, b1, d_a1, n1w, n1b, add=dskip, colsum_of=ctx.db_param)
        return dx, dn1w, dn1b, dw1, dn2w, dn2b, dw2, dwsc, None, None, None, None


def res_block(x, n1w, n1b, c1w, n2w, n2b, c2w, scw=None, groups: int = 32, eps: float = 1e-6, pool: bool = False,
              next_gn: int = 0):
    """pool: also apply the 2x2 average pool that follows the block (the encoder's Downsample, autoencoder.py:89-91);
    next_gn: a GroupNorm with that many groups is the NEXT consumer of the result (its sums ride in conv2's drain)"""
    return ResBlockFn.apply(x, n1w, n1b, c1w, n2w, n2b, c2w, scw, groups, eps, pool, next_gn)


def group_norm_silu(x, weight, bias, groups: int = 32, eps: float = 1e-6, silu: bool = True):
    return GroupNormSiLUFn.apply(x, weight, bias, groups, eps, silu)


class AvgPool2x2Fn(torch.autograd.Function):
    """autoencoder.py:89-91"""

    @staticmethod
    def forward(ctx, x):
        _require_gpu(x)
        return raw_pool(nhwc(x), 0.25)

    @staticmethod
    def backward(ctx, dy):
        return raw_unpool(nhwc(dy), 0.25)


def avg_pool2x2(x):
    return AvgPool2x2Fn.apply(x)


def mse_loss(recon, target, true_channels: int | None = None):
    """mean((recon - target)^2)  (vqvae/model.py:137,272); target carries no gradient.
    ``true_channels``: logical channel count when both tensors carry zero-padded channels (the mean is
    taken over the un-padded element count)."""
    n, c, h, w = recon.shape
    return _MSE.apply(recon, target, float(n * (true_channels or c) * h * w))


class _MSE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, recon, target, denom: float):
        _require_gpu(recon)
        recon = nhwc(recon)
        target = nhwc(target.to(torch.float32))
        if recon.shape != target.shape:
            raise RuntimeError(f'vqk: mse shapes differ {tuple(recon.shape)} vs {tuple(target.shape)}')
        sse = torch.zeros((), dtype=torch.float32, device=recon.device)
        _native.check(_native.lib().vqk_sse(dcode(recon.dtype), recon.data_ptr(), target.data_ptr(), recon.numel(),
                                            sse.data_ptr(), _stream()), 'sse')
        ctx.save_for_backward(recon, target)
        ctx.denom = denom
        return sse / denom

    @staticmethod
    def backward(ctx, dloss):
        recon, target = ctx.saved_tensors
        d = torch.empty_like(recon, memory_format=_CL)
        gs = dloss.to(torch.float32).contiguous()
        _native.check(_native.lib().vqk_mse_tanh_backward(dcode(recon.dtype), recon.data_ptr(), target.data_ptr(),
                                                          recon.numel(), 1.0 / ctx.denom, gs.data_ptr(), 0,
                                                          d.data_ptr(), _stream()), 'mse_backward')
        return d, None, None


# ------------------------------------------------------------------------------------------------------
# switches of the operator families that live in _ops_vq.py / _ops_gan.py (kept HERE: tests and tools flip them as `ops.X = ...`)
# ------------------------------------------------------------------------------------------------------
VQ_FILTER = _native.switch('VQK_VQ_FILTER', '1') != '0'
VQ_FUSED = _native.switch('VQK_VQ_FUSED', '1') != '0'      # one forward kernel + one backward kernel (0: the round-3 launch sequence)
ENTROPY_FUSED_ROWS = _native.switch('VQK_ENTROPY_FUSED_ROWS', '1') != '0'
ENTROPY_SPLIT_GEMM = _native.switch('VQK_ENTROPY_SPLIT_GEMM', '1') != '0'     # bf16 compute mode: the entropy cotangent's two GEMMs as bf16 split products
ACT_CODE = {'linear': 0, 'tanh': 1, 'relu': 2, 'lrelu': 3}
FUSE_DISC_BLOCK = _native.switch('VQK_FUSE_DISC_BLOCK', '1') != '0'
DIRECT_LINEAR_WGRAD = _native.switch('VQK_DIRECT_LINEAR_WGRAD', '1') != '0'   # linear ConvActFn layers: scaled weight gradient straight into the arena
DIRECT_BIAS_GRAD = _native.switch('VQK_DIRECT_BIAS_GRAD', '1') != '0'    # ConvActFn: bias gradient straight into the optimizer's arena


# ------------------------------------------------------------------------------------------------------
# tracing: named host ranges around every macro-op (VQK_TRACE=1)
# ------------------------------------------------------------------------------------------------------
# The reference names its custom ops for the profiler (stylegan2_discriminator/utils/misc.py:104-110: profiled_function ->
# torch.autograd.profiler.record_function); here every autograd Function of this file gets a roctx range "vqk::<op>.forward /
# .backward" (torch.cuda.nvtx = roctx on ROCm; visible to `rocprofv3 --marker-trace` and to torch.profiler) when VQK_TRACE=1.
# Off by default: nothing is wrapped, no per-call cost.  The ranges are HOST ranges around the launches -- a replayed hipGraph
# carries none (tools/trace_step.sh traces eager steps).
TRACE = _native.switch('VQK_TRACE', '0') == '1'


class trace_range:
    """``with ops.trace_range('name'):`` -- a roctx range when tracing is on, nothing otherwise"""

    def __init__(self, name: str):
        self.name = name

    def __enter__(self):
        if TRACE:
            torch.cuda.nvtx.range_push(self.name)

    def __exit__(self, *exc):
        if TRACE:
            torch.cuda.nvtx.range_pop()


def _ranged(name: str, fn):
    import functools

    @functools.wraps(fn)
    def wrapped(*args, **kwargs):
        torch.cuda.nvtx.range_push(name)
        try:
            return fn(*args, **kwargs)
        finally:
            torch.cuda.nvtx.range_pop()
    return wrapped


def install_tracing() -> int:
    """wrap forward / backward of every autograd Function defined here in a named range; returns the number of ranges installed"""
    n = 0
    for cls_name, obj in list(globals().items()):
        if isinstance(obj, type) and issubclass(obj, torch.autograd.Function) and obj is not torch.autograd.Function:
            label = cls_name.lstrip('_')
            label = label[:-2] if label.endswith('Fn') else label
            for meth in ('forward', 'backward'):
                sm = obj.__dict__.get(meth)
                if isinstance(sm, staticmethod) and not getattr(sm.__func__, '_vqk_ranged', False):
                    w = _ranged(f'vqk::{label}.{meth}', sm.__func__)
                    w._vqk_ranged = True
                    setattr(obj, meth, staticmethod(w))
                    n += 1
    return n


if TRACE:
    install_tracing()


# ------------------------------------------------------------------------------------------------------
# operator families kept in their own files (this module stays the one import: `ops.VQLookupFn`, `ops.conv_act`, ...)
# ------------------------------------------------------------------------------------------------------
from ._ops_vq import *        # noqa: E402,F401,F403  quantizers
from ._ops_gan import *       # noqa: E402,F401,F403  VQ-GAN loss path, the reference's two plugins

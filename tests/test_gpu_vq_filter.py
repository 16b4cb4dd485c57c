"""The bf16 candidate filter + exact fp32 re-rank form of the nearest-codeword assignment (csrc/vq_filter.hip; reference:
vqvae/modules/vector_quantizers.py:37-44 / :337-343 followed by torch.argmin) must return EXACTLY the indices of the
exact-fp32 MFMA kernel it replaces and of the C oracle -- on well-separated data, on near-ties, on exact ties, on a
collapsed codebook (candidate-list overflow -> in-kernel exact fallback), with K = 8192 (tiles beyond the LDS cache are
recomputed) and with ragged N."""
import importlib

import numpy as np
import pytest
import torch

from oracle import vq_c

pytestmark = pytest.mark.gpu

ops = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd.ops')
native = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd._native')
DEV = 'cuda:0'


def _both(z, e, assoc):
    zd, ed = z.to(DEV).contiguous(), e.to(DEV).contiguous()
    assert ops.VQ_FILTER
    calls = {'n': 0}
    lib = native.lib()
    real = lib.vqk_vq_assign_filtered_f32

    class Counting:
        def __getattr__(self, name):
            if name == 'vqk_vq_assign_filtered_f32':
                def f(*a):
                    calls['n'] += 1
                    return real(*a)
                return f
            return getattr(lib, name)
    saved = native.lib
    native.lib = lambda: Counting()
    try:
        got = ops.vq_assign(zd, ed, assoc).cpu().numpy()
    finally:
        native.lib = saved
    assert calls['n'] == 1                                           # the filter path really ran
    ops.VQ_FILTER = False
    try:
        want = ops.vq_assign(zd, ed, assoc).cpu().numpy()
    finally:
        ops.VQ_FILTER = True
    return got, want


def _trained_like(n, k, g, noise=0.01):
    z = torch.randn(n, 256, generator=g) * 0.36
    e = z[torch.randperm(n, generator=g)[:k]] + noise * torch.randn(k, 256, generator=g)
    return z, e


@pytest.mark.parametrize('assoc', [0, 1])
@pytest.mark.parametrize('case', ['random', 'trained', 'near_ties', 'exact_ties', 'collapsed', 'scaled_small', 'ragged', 'k8192', 'mixed_norms'])
def test_filter_indices_equal_exact_kernel_and_oracle(case, assoc):
    g = torch.Generator().manual_seed(sum(map(ord, case)) + assoc)
    n, k = 8192, 1024
    if case == 'random':                         # uniform tiny codebook against N(0,1) latents: distances bunch up
        z = torch.randn(n, 256, generator=g)
        e = (torch.rand(k, 256, generator=g) * 2 - 1) / k
    elif case == 'trained':
        z, e = _trained_like(n, k, g)
    elif case == 'near_ties':                    # codes differing in the last bits: far inside the bf16 filter's margin
        z, e = _trained_like(n, k, g)
        e[1::2] = e[0::2] * (1 + 2.0 ** -20)
    elif case == 'exact_ties':                   # duplicated codes: the lower index must win
        z, e = _trained_like(n, k, g)
        e[k // 2:] = e[:k // 2]
    elif case == 'collapsed':                    # every code (nearly) the same vector: thousands of candidates per block
        z = torch.randn(n, 256, generator=g)
        e = torch.randn(1, 256, generator=g).repeat(k, 1) + 1e-6 * torch.randn(k, 256, generator=g)
    elif case == 'scaled_small':
        z, e = _trained_like(n, k, g)
        z, e = z * 1e-3, e * 1e-3
    elif case == 'ragged':
        n = 1000 + 17
        z, e = _trained_like(4096, k, g)
        z = z[:n].contiguous()
    elif case == 'k8192':
        n, k = 4096, 8192
        z = torch.randn(n, 256, generator=g) * 0.36
        e = torch.randn(k, 256, generator=g) * 0.36
    else:                                        # code norms spread over four decades
        z = torch.randn(n, 256, generator=g)
        e = torch.randn(k, 256, generator=g) * torch.logspace(-2, 2, k).unsqueeze(1)
    got, want = _both(z, e, assoc)
    assert np.array_equal(got, want), int((got != want).sum())
    rows = slice(0, 512)                          # the scalar C oracle is ~0.5 ms per (row, 1024 codes)
    ref, _, _, _ = vq_c.assign(z[rows].numpy(), e.numpy(), assoc)
    assert np.array_equal(got[rows], ref)
    if case == 'exact_ties':
        assert (got < k // 2).all()


def _cases(case, g):
    n, k = 8192, 1024
    if case == 'trained':
        z, e = _trained_like(n, k, g)
    elif case == 'collapsed':
        z = torch.randn(n, 256, generator=g)
        e = torch.randn(1, 256, generator=g).repeat(k, 1) + 1e-6 * torch.randn(k, 256, generator=g)
    elif case == 'few_codes':                    # a fresh model: every row lands on a handful of codes (no list overflow)
        z = torch.randn(n, 256, generator=g) * 0.5
        e = (torch.rand(k, 256, generator=g) * 2 - 1) / k
        e[:3] = z[:3] * 0.9
    elif case == 'ragged':
        n = 2000 + 13
        z, e = _trained_like(4096, k, g)
        z = z[:n].contiguous()
    else:                                        # k8192
        n, k = 4096, 8192
        z = torch.randn(n, 256, generator=g) * 0.36
        e = torch.randn(k, 256, generator=g) * 0.36
    return z, e, n, k


@pytest.mark.parametrize('assoc', [0, 1])
@pytest.mark.parametrize('case', ['trained', 'collapsed', 'few_codes', 'ragged', 'k8192'])
def test_fused_forward_kernel_equals_separate_launches(case, assoc):
    """vqk_vq_forward_f32 (round 4: |z|^2 + filter + re-rank + gather + sum (q - z)^2 + histogram in ONE kernel, codebook
    derivatives prepared once by vqk_vq_prepare_f32; reference: vector_quantizers.py:37-56) against the separate launches
    it replaces (row norms, exact-fp32 assignment, vqk_vq_gather_f32) and the C oracle: indices and q bit-identical, the
    histogram equal, the loss sum to fp32 summation-order accuracy."""
    g = torch.Generator().manual_seed(sum(map(ord, case)) + 7 * assoc)
    z, e, n, k = _cases(case, g)
    zd, ed = z.to(DEV).contiguous(), e.to(DEV).contiguous()
    lib = native.lib()
    s = torch.cuda.current_stream().cuda_stream
    ws = torch.empty(lib.vqk_vq_filter_ws_bytes(k, 256), dtype=torch.uint8, device=DEV)
    native.check(lib.vqk_vq_prepare_f32(ed.data_ptr(), k, 256, ws.data_ptr(), ws.numel(), s), 'prepare')
    idx = torch.empty(n, dtype=torch.int64, device=DEV)
    q32 = torch.empty(n, 256, device=DEV)
    qlo = torch.empty(n, 256, dtype=torch.bfloat16, device=DEV)
    sse = torch.zeros((), device=DEV)
    hist = torch.zeros(k, dtype=torch.int32, device=DEV)
    native.check(lib.vqk_vq_forward_f32(zd.data_ptr(), ed.data_ptr(), ws.data_ptr(), ws.numel(), n, k, 256, assoc, idx.data_ptr(),
                                        q32.data_ptr(), qlo.data_ptr(), sse.data_ptr(), hist.data_ptr(), s), 'forward')
    ops.VQ_FILTER = False
    try:
        want = ops.vq_assign(zd, ed, assoc)
    finally:
        ops.VQ_FILTER = True
    assert torch.equal(idx, want), int((idx != want).sum())
    rows = slice(0, 256)
    ref, _, _, _ = vq_c.assign(z[rows].numpy(), e.numpy(), assoc)
    assert np.array_equal(idx[rows].cpu().numpy(), ref)
    q_w = torch.empty(n, 256, device=DEV); ql_w = torch.empty(n, 256, dtype=torch.bfloat16, device=DEV)
    sse_w = torch.zeros((), device=DEV); hist_w = torch.zeros(k, dtype=torch.int32, device=DEV)
    native.check(lib.vqk_vq_gather_f32(zd.data_ptr(), ed.data_ptr(), want.data_ptr(), n, k, 256, q_w.data_ptr(), ql_w.data_ptr(),
                                       sse_w.data_ptr(), hist_w.data_ptr(), s), 'gather')
    assert torch.equal(q32, q_w) and torch.equal(qlo, ql_w) and torch.equal(hist, hist_w)
    assert int(hist.sum()) == n
    exact = float(((ed[want].double() - zd.double()) ** 2).sum())
    assert abs(float(sse) - exact) <= 1e-5 * exact and abs(float(sse_w) - exact) <= 1e-5 * exact
    # the fp32 copy of q is optional (bf16 mode writes q_lo only); nothing else may change
    idx2 = torch.empty_like(idx); ql2 = torch.empty_like(qlo)
    native.check(lib.vqk_vq_forward_f32(zd.data_ptr(), ed.data_ptr(), ws.data_ptr(), ws.numel(), n, k, 256, assoc, idx2.data_ptr(),
                                        0, ql2.data_ptr(), 0, 0, s), 'forward')
    assert torch.equal(idx2, idx) and torch.equal(ql2, qlo)


@pytest.mark.parametrize('dq_dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('case', ['trained', 'few_codes', 'ragged'])
def test_fused_backward_kernel_equals_two_kernel_form(case, dq_dtype):
    """vqk_vq_backward_fused_f32 (one kernel: dz + per-block code sums in LDS + one atomic row per distinct code) against
    vqk_vq_backward_f32 (vq_backward_kernel + vq_code_grad_kernel) and an fp64 evaluation of vector_quantizers.py:52-56"""
    g = torch.Generator().manual_seed(sum(map(ord, case)))
    z, e, n, k = _cases(case, g)
    zd, ed = z.to(DEV).contiguous(), e.to(DEV).contiguous()
    idx = ops.vq_assign(zd, ed, 0)
    dq = torch.randn(n, 256, generator=g).to(dq_dtype).to(DEV)
    gs = torch.tensor(0.7, device=DEV)
    lib, s = native.lib(), torch.cuda.current_stream().cuda_stream
    cz, ce = 0.25 * 2.0 / (n * 256), 2.0 / (n * 256)
    out = {}
    for name in ('vqk_vq_backward_fused_f32', 'vqk_vq_backward_f32'):
        dz = torch.empty(n, 256, device=DEV); de = torch.zeros(k, 256, device=DEV)
        native.check(getattr(lib, name)(zd.data_ptr(), ed.data_ptr(), idx.data_ptr(), dq.data_ptr(), ops.dcode(dq_dtype), n, k, 256,
                                        cz, ce, gs.data_ptr(), dz.data_ptr(), de.data_ptr(), s), name)
        out[name] = (dz, de)
    assert torch.equal(out['vqk_vq_backward_fused_f32'][0], out['vqk_vq_backward_f32'][0])        # dz: same arithmetic per element
    q = ed[idx].double()
    de_ref = torch.zeros(k, 256, dtype=torch.float64, device=DEV).index_add_(0, idx, 0.7 * ce * (q - zd.double()))
    for name, (dz, de) in out.items():
        err = float((de.double() - de_ref).norm() / de_ref.norm())
        assert err < 2e-6, (name, err)
    dz_ref = dq.double() + 0.7 * cz * (zd.double() - q)
    assert float((out['vqk_vq_backward_fused_f32'][0].double() - dz_ref).abs().max()) < 1e-6 * float(dz_ref.abs().max()) + 1e-9
    # dz only (EMA quantizer: the codebook has no gradient)
    dz = torch.empty(n, 256, device=DEV)
    native.check(lib.vqk_vq_backward_fused_f32(zd.data_ptr(), ed.data_ptr(), idx.data_ptr(), dq.data_ptr(), ops.dcode(dq_dtype), n, k,
                                               256, cz, ce, gs.data_ptr(), dz.data_ptr(), 0, s), 'fused, no de')
    assert torch.equal(dz, out['vqk_vq_backward_fused_f32'][0])


def test_prepared_codebook_follows_the_codebook():
    """the forward kernel reads DERIVED data (bf16 copy, |e|^2, margins) prepared when the codebook changes: a torch in-place
    write, the EMA update through the C-ABI and an optimizer step must each be seen by the next lookup"""
    g = torch.Generator().manual_seed(11)
    z, e, n, k = _cases('trained', g)
    zd = z.to(DEV).view(8, 32, 32, 256).permute(0, 3, 1, 2)             # [B, D, H, W] logical, NHWC memory
    cb = torch.nn.Parameter(e.to(DEV).contiguous())

    def lookup():
        with torch.no_grad():
            _, idx, _, _ = ops.VQLookupFn.apply(zd, cb, 0.25, True, 0, torch.float32)
        return idx.reshape(-1)

    def exact():
        ops.VQ_FILTER = False
        try:
            return ops.vq_assign(z.to(DEV).contiguous(), cb.detach().contiguous(), 0)
        finally:
            ops.VQ_FILTER = True
    assert torch.equal(lookup(), exact())
    with torch.no_grad():
        cb.copy_(cb.roll(5, 0))                                       # version bump
    assert torch.equal(lookup(), exact())
    stats = ops.ema_stats(z.to(DEV).contiguous(), lookup(), k)
    cnt = torch.ones(k, device=DEV); wgt = cb.detach().clone()
    ops.ema_apply(stats, cnt, wgt, cb.data, 0.5, 1e-5, 8.0)          # raw pointer write: ema_apply refreshes the entry
    assert torch.equal(lookup(), exact())


@pytest.mark.parametrize('case', ['trained', 'few_codes', 'ragged'])
def test_fused_ema_statistics_equal_the_per_element_form(case):
    """vqk_ema_stats_fused_f32 (rows of a block that share a code summed in LDS, one atomic row per distinct code) against
    vqk_ema_stats_f32 and an fp64 evaluation of vector_quantizers.py:159-163 (counts, dw = one_hot^T z)"""
    g = torch.Generator().manual_seed(sum(map(ord, case)) + 1)
    z, e, n, k = _cases(case, g)
    zd, ed = z.to(DEV).contiguous(), e.to(DEV).contiguous()
    idx = ops.vq_assign(zd, ed, 0)
    lib, s = native.lib(), torch.cuda.current_stream().cuda_stream
    out = {}
    for name in ('vqk_ema_stats_fused_f32', 'vqk_ema_stats_f32'):
        buf = torch.zeros(k + k * 256, device=DEV)
        native.check(getattr(lib, name)(zd.data_ptr(), idx.data_ptr(), n, k, 256, buf.data_ptr(), buf[k:].data_ptr(), s), name)
        out[name] = buf
    counts_ref = torch.bincount(idx, minlength=k).double()
    dw_ref = torch.zeros(k, 256, dtype=torch.float64, device=DEV).index_add_(0, idx, zd.double())
    for name, buf in out.items():
        assert torch.equal(buf[:k].double(), counts_ref), name
        err = float((buf[k:].view(k, 256).double() - dw_ref).norm() / dw_ref.norm())
        assert err < 2e-6, (name, err)

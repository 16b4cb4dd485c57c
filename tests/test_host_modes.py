"""Host-side logic that needs no GPU: the compute-mode resolution of `VQVAE(compute_dtype=...)` and the per-backward-call gradient
modes of ops.py (`no_param_grads` / `no_direct_grad` travel with the autograd graph task of the call they wrap, not in a
process-wide flag -- VERDICT r5 item 7)."""
import importlib
import threading

import pytest
import torch

PKG = 'vqvae-vqgan-pytorch-lightning_amd'
ops = importlib.import_module(PKG + '.ops')
ae = importlib.import_module(PKG + '.modules.autoencoder')


def test_resolve_compute_dtype():
    assert ae.resolve_compute_dtype(torch.float32) == (torch.float32, 'fp32')
    assert ae.resolve_compute_dtype(torch.bfloat16) == (torch.bfloat16, 'fp32')
    assert ae.resolve_compute_dtype('bf16x3') == (torch.float32, 'bf16x3')
    with pytest.raises(ValueError):
        ae.resolve_compute_dtype(torch.float16)
    enc = ae.Encoder(32, 1, (1, 2), 16)
    ae.set_compute_dtype(enc, 'bf16x3')
    assert enc.compute_dtype == torch.float32 and enc.conv_products == 'bf16x3'
    with pytest.raises(ValueError):
        ops.set_conv_products('tf32')


def test_x3_eligibility_rules():
    f32, bf = torch.float32, torch.bfloat16
    assert ops.x3_serves(f32, None, 64, 64, 128, 128, 3) and ops.x3_serves(f32, f32, 16, 16, 512, 256, 1)
    assert not ops.x3_serves(bf, None, 64, 64, 128, 128, 3)          # the bf16 mode has its own kernels
    assert not ops.x3_serves(f32, None, 64, 64, 4, 128, 3)            # the 3-channel edge convs: exact fp32 (conv_thin_f32.hip)
    assert not ops.x3_serves(f32, None, 8, 8, 128, 128, 3)            # 8-wide maps: the 8x16-pixel tile does not fit
    assert not ops.x3_serves(f32, bf, 64, 64, 128, 128, 3)


class _Probe(torch.autograd.Function):
    seen = []

    @staticmethod
    def forward(ctx, x, tag):
        ctx.tag = tag
        return x * 2.0

    @staticmethod
    def backward(ctx, g):
        _Probe.seen.append((ctx.tag, ops._grad_modes(), threading.get_ident()))
        return g * 2.0, None


def test_gradient_modes_travel_with_the_backward_call():
    _Probe.seen.clear()
    x = torch.ones(3, requires_grad=True)
    y = _Probe.apply(x, 'plain').sum()
    y.backward()
    with ops.no_param_grads():
        z = _Probe.apply(x, 'noparam').sum()
        ops.backward(z)
        with ops.no_direct_grad():
            g, = ops.autograd_grad(_Probe.apply(x, 'both').sum(), x)
        assert ops._grad_modes() == (True, False)                 # (the calling thread's own view inside the context)
    u = _Probe.apply(x, 'after').sum()
    u.backward()
    modes = {t: m for t, m, _ in _Probe.seen}
    assert modes == {'plain': (True, True), 'noparam': (True, False), 'both': (False, False), 'after': (True, True)}
    assert not ops._TASK_MODES and ops._grad_modes() == (True, True)
    assert torch.equal(g, torch.full((3,), 2.0))


def test_gradient_modes_of_two_threads_do_not_mix():
    """thread A holds `no_param_grads` open while thread B runs plain backwards (and the other way round): with the rounds-1-5
    save / restore globals B would have skipped its parameter gradients, or an interleaved exit would have left the flag stuck"""
    _Probe.seen.clear()
    a_in, b_go, a_done = threading.Event(), threading.Event(), threading.Event()
    errors = []

    def thread_a():
        try:
            x = torch.ones(2, requires_grad=True)
            with ops.no_param_grads():
                a_in.set()
                b_go.wait(10)
                for _ in range(20):
                    ops.backward(_Probe.apply(x, 'A').sum())
        except Exception as exc:                                 # noqa: BLE001
            errors.append(exc)
        finally:
            a_done.set()

    def thread_b():
        try:
            x = torch.ones(2, requires_grad=True)
            a_in.wait(10)
            b_go.set()
            for _ in range(20):
                _Probe.apply(x, 'B').sum().backward()
                assert ops._grad_modes() == (True, True)
        except Exception as exc:                                 # noqa: BLE001
            errors.append(exc)

    ta, tb = threading.Thread(target=thread_a), threading.Thread(target=thread_b)
    ta.start(); tb.start(); ta.join(30); tb.join(30)
    assert not errors, errors
    assert all(m == ((True, False) if t == 'A' else (True, True)) for t, m, _ in _Probe.seen)
    assert sum(1 for t, _, _ in _Probe.seen if t == 'A') == 20 and sum(1 for t, _, _ in _Probe.seen if t == 'B') == 20
    assert not ops._TASK_MODES

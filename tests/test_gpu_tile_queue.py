"""Dynamic tile queue of the persistent matrix/auxiliary-wave conv kernel (csrc/conv_mx.hip, include/vqk.h: vqk_set_tile_queue,
tuning slot TILE_QUEUE).  The reference leaves block scheduling to cuDNN (vqvae/modules/autoencoder.py:57-60 -> F.conv2d); what
has to hold here is that the queue changes WHO computes a tile, never WHAT is computed: outputs bit-identical to the static
share in both queue modes, alone and while another kernel holds CUs, and the queue words zero again after every launch."""
import importlib

import pytest
import torch

pytestmark = pytest.mark.gpu

ops = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd.ops')
native = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd._native')
DEV, BF, CL = 'cuda:0', torch.bfloat16, torch.channels_last


def _queue_words():
    ops._stream()
    dev = torch.cuda.current_device()
    return ops._TILE_QUEUE[ops._wkey()]


def _set_mode(m):
    native.check(native.lib().vqk_set_tuning(b'TILE_QUEUE', m), 'set_tuning')


@pytest.fixture(autouse=True)
def _reset_mode():
    yield
    native.lib().vqk_reset_tuning()
    native.apply_env_tuning()


# n, cin, cout, h, w, bias, residual, pool   -- every case has more 256-pixel tiles than the chip has CUs
CASES = [(8, 128, 128, 128, 128, 0, 0, 0), (8, 128, 128, 128, 128, 1, 1, 0), (2, 128, 128, 256, 256, 0, 1, 0),
         (16, 256, 256, 64, 64, 1, 0, 0), (12, 128, 256, 64, 64, 0, 1, 1), (5, 256, 128, 96, 96, 0, 0, 0),
         (3, 64, 128, 160, 160, 1, 0, 0),          # 64 input channels: tiles of two units (the ring is pre-fetched)
         (9, 128, 128, 48, 112, 0, 0, 0)]


@pytest.mark.parametrize('n,cin,cout,h,w,hb,hr,pool', CASES)
def test_queue_modes_bit_identical_to_static_share(n, cin, cout, h, w, hb, hr, pool):
    g = torch.Generator(device=DEV).manual_seed(n + cin + cout + h + hb + 2 * hr + 4 * pool)
    x = torch.randn(n, cin, h, w, device=DEV, generator=g).to(BF).contiguous(memory_format=CL)
    wt = (torch.randn(cout, 3, 3, cin, device=DEV, generator=g) / (3 * cin ** 0.5)).reshape(-1)
    bias = torch.randn(cout, device=DEV, generator=g) if hb else None
    res = torch.randn(n, cout, h, w, device=DEV, generator=g).to(BF).contiguous(memory_format=CL) if hr else None
    assert ops.weight_layout(BF, n, h, w, cin, cout, 3, False) == 1
    wq = ops.pack_weights(wt, BF, cout, cin, 3, False, 1)
    words = _queue_words()
    outs = {}
    for mode in (0, 1, 2):
        _set_mode(mode)
        for rep in range(3):
            if pool:
                y = ops.raw_conv_fprop_pooled(x, wq, bias, res, 3, False, cout, 0.25)
            else:
                y = ops.raw_conv_fprop(x, wq, bias, res, 3, False, 0, BF, cout, 1)
            torch.cuda.synchronize()
            assert int(words.abs().sum()) == 0, (mode, rep, words[:9].tolist())
            if mode in outs:
                assert torch.equal(outs[mode], y)
            outs[mode] = y
    assert torch.equal(outs[0], outs[1])
    assert torch.equal(outs[0], outs[2])


def test_queue_with_other_kernel_forms():
    """1x1 tiles (no halo), the phase-form upsample conv (forward: phase = a tile dimension, data gradient: phase = a unit
    dimension) and the GroupNorm sums in the drain"""
    g = torch.Generator(device=DEV).manual_seed(5)
    n, c = 8, 128
    x = torch.randn(n, c, 64, 64, device=DEV, generator=g).to(BF).contiguous(memory_format=CL)
    w3 = torch.nn.Parameter((torch.randn(c, c, 3, 3, device=DEV, generator=g) / 34).contiguous(memory_format=CL))
    w1 = (torch.randn(256, 1, 1, c, device=DEV, generator=g) / 11).reshape(-1)
    bias = torch.randn(c, device=DEV, generator=g)
    xl = torch.randn(n, c, 128, 128, device=DEV, generator=g).to(BF).contiguous(memory_format=CL)
    words = _queue_words()
    res = {}
    for mode in (0, 1, 2):
        _set_mode(mode)
        ops.clear_pack_cache()
        up = ops.raw_conv_ups_phase(x, ops.packed_weight(w3, c, c, BF, 3, False, 2), bias, c, False)
        dn = ops.raw_conv_ups_phase(xl, ops.packed_weight(w3, c, c, BF, 3, True, 2), None, c, True)
        assert up is not None and dn is not None
        l1 = ops.weight_layout(BF, n, 128, 128, c, 256, 1, False)
        y1 = ops.raw_conv_fprop(xl, ops.pack_weights(w1, BF, 256, c, 1, False, l1), None, None, 1, False, 0, BF, 256, l1)
        ops.set_deterministic(True)            # per-tile slots: the sums are bit-reproducible whoever computes the tile
        try:
            ys = ops.raw_conv_fprop_gnstats(xl, ops.packed_weight(w3, c, c, BF, 3, False, 1), None, None, False, c, 32)
            assert ys is not None
            gw, gb = torch.ones(c, device=DEV), torch.zeros(c, device=DEV)
            gy, st = ops.raw_gn_forward(ys, gw, gb, 32, 1e-6, True, presummed=True)
        finally:
            ops.set_deterministic(False)
        torch.cuda.synchronize()
        assert int(words.abs().sum()) == 0, mode
        res[mode] = (up, dn, y1, ys, gy, st)
    for mode in (1, 2):
        for a, b in zip(res[0], res[mode]):
            assert torch.equal(a, b), mode


def test_queue_under_contention_and_in_graph_replay():
    """a stand-in for a collective's kernel (vqk_probe_stream_add: persistent blocks on a side stream) holds CUs while the conv
    runs: late blocks find less work -- same result; the same launches replayed from a hipGraph leave the words zero too"""
    g = torch.Generator(device=DEV).manual_seed(9)
    n, c, h = 8, 128, 128
    x = torch.randn(n, c, h, h, device=DEV, generator=g).to(BF).contiguous(memory_format=CL)
    wt = (torch.randn(c, 3, 3, c, device=DEV, generator=g) / 34).reshape(-1)
    wq = ops.pack_weights(wt, BF, c, c, 3, False, 1)
    lib = native.lib()
    _set_mode(0)
    want = ops.raw_conv_fprop(x, wq, None, None, 3, False, 0, BF, c, 1)
    src = torch.zeros(8 << 20, device=DEV)
    dst = torch.zeros(8 << 20, device=DEV)
    side = torch.cuda.Stream()
    words = _queue_words()
    for mode in (1, 2):
        _set_mode(mode)
        for blocks in (16, 64, 200):
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                native.check(lib.vqk_probe_stream_add(src.data_ptr(), dst.data_ptr(), src.numel() * 4, blocks, 2, 1, side.cuda_stream), 'probe')
            ys = [ops.raw_conv_fprop(x, wq, None, None, 3, False, 0, BF, c, 1) for _ in range(4)]
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            for y in ys:
                assert torch.equal(y, want), (mode, blocks)
            assert int(words.abs().sum()) == 0, (mode, blocks)
    # graph replay: the queue words are part of the captured kernel arguments and must be zero before / after every replay
    _set_mode(2)
    cap = torch.cuda.Stream()
    cap.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(cap):
        ops._stream()
        ops.raw_conv_fprop(x, wq, None, None, 3, False, 0, BF, c, 1)
        cap_words = ops._TILE_QUEUE[ops._wkey()]
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=cap):
        y1 = ops.raw_conv_fprop(x, wq, None, None, 3, False, 0, BF, c, 1)
        y2 = ops.raw_conv_fprop(y1, wq, None, None, 3, False, 0, BF, c, 1)
    _set_mode(0)
    want2 = ops.raw_conv_fprop(want, wq, None, None, 3, False, 0, BF, c, 1)
    for _ in range(3):
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(y1, want) and torch.equal(y2, want2)
        assert int(cap_words.abs().sum()) == 0

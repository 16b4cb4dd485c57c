"""GroupNorm statistics fused into the drain of the matrix/auxiliary-wave 3x3 kernel (vqk_conv2d_fprop_gnstats) against the
separate statistics pass it replaces (vqk_gn_forward), which the golden tests pin to the reference's GroupNorm
(autoencoder.py:25-39): same conv output bit for bit, (mean, rstd) equal to fp32 rounding, workspace left zero."""
import importlib

import pytest
import torch

pytestmark = pytest.mark.gpu

ops = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd.ops')
DEV, BF, CL = 'cuda:0', torch.bfloat16, torch.channels_last

# n, cin, cout, h, w, ups, bias, residual, pool   (every case has >= 64 tiles: served by the fused kernel; the short ones
# give a block ONE tile, or some blocks one and some two)
CASES = [(4, 128, 128, 128, 128, 0, 0, 0, 0), (2, 128, 128, 128, 128, 0, 0, 1, 0), (4, 256, 256, 64, 64, 0, 0, 1, 0),
         (3, 128, 128, 128, 128, 0, 1, 0, 0), (5, 128, 256, 64, 64, 0, 0, 0, 0), (4, 128, 128, 128, 128, 0, 0, 1, 1),
         (4, 256, 256, 128, 128, 0, 0, 1, 1), (4, 128, 256, 128, 128, 0, 0, 0, 1), (2, 128, 512, 128, 128, 0, 1, 1, 1),
         (8, 128, 128, 128, 128, 0, 0, 0, 0), (4, 64, 256, 128, 128, 0, 1, 1, 0), (2, 64, 512, 128, 128, 0, 0, 1, 0),
         (8, 128, 128, 128, 128, 0, 0, 1, 1), (8, 128, 128, 64, 64, 1, 1, 0, 0), (6, 128, 256, 96, 128, 0, 0, 0, 0)]


@pytest.mark.parametrize('n,cin,cout,h,w,ups,hb,hr,pool', CASES)
def test_fused_stats_match_separate_pass(n, cin, cout, h, w, ups, hb, hr, pool):
    g = torch.Generator(device=DEV).manual_seed(cin + cout + h + 3 * ups + hb + 2 * hr + 4 * pool)
    x = (torch.randn(n, cin, h, w, device=DEV, generator=g) + 0.3).to(BF).contiguous(memory_format=CL)
    wt = (torch.randn(cout, 3, 3, cin, device=DEV, generator=g) / (3 * cin ** 0.5)).reshape(-1)
    s = 2 if ups else 1
    ho, wo = h * s, w * s
    bias = torch.randn(cout, device=DEV, generator=g) if hb else None
    res = torch.randn(n, cout, ho, wo, device=DEV, generator=g).to(BF).contiguous(memory_format=CL) if hr else None
    layout = ops.weight_layout(BF, n, h, w, cin, cout, 3, bool(ups))
    assert layout == 1
    wq = ops.pack_weights(wt, BF, cout, cin, 3, False, layout)
    gw = torch.randn(cout, device=DEV, generator=g)
    gb = torch.randn(cout, device=DEV, generator=g)
    if pool:
        y_ref = ops.raw_conv_fprop_pooled(x, wq, bias, res, 3, bool(ups), cout, 0.25)
    else:
        y_ref = ops.raw_conv_fprop(x, wq, bias, res, 3, bool(ups), 0, BF, cout, layout)
    a_ref, st_ref = ops.raw_gn_forward(y_ref, gw, gb, 32, 1e-6, True)
    y = ops.raw_conv_fprop_gnstats(x, wq, bias, res, bool(ups), cout, 32, pool=bool(pool), pool_scale=0.25)
    assert y is not None, 'the fused kernel must serve this shape'
    a, st = ops.raw_gn_forward(y, gw, gb, 32, 1e-6, True, presummed=True)
    torch.cuda.synchronize()
    assert torch.equal(y, y_ref)
    # mean / rstd: the same sums in a different order (fp32 partials per tile, fp64 across tiles)
    torch.testing.assert_close(st.view(-1, 2)[:, 0], st_ref.view(-1, 2)[:, 0], rtol=0, atol=2e-5)
    torch.testing.assert_close(st.view(-1, 2)[:, 1], st_ref.view(-1, 2)[:, 1], rtol=2e-5, atol=0)
    # the output may differ by ONE bf16 step where the fp32 value sits on a rounding boundary
    assert float(((a.float() - a_ref.float()).abs() / (a_ref.float().abs() + 1.0)).max()) <= 2.0 ** -7
    assert float((a.float() - a_ref.float()).norm() / a_ref.float().norm()) < 1e-4
    ws = ops._gn_ws(x.device, n * 32 * 2 + n)
    assert float(ws.abs().max()) == 0.0, 'the GroupNorm workspace must be left zero'


@pytest.mark.parametrize('n,h,w', [(4, 128, 128), (32, 64, 64), (3, 96, 64), (2, 256, 256)])
def test_thin_in_conv_leaves_the_sums_of_its_output(n, h, w):
    """the encoder's first conv (padded 3-channel image -> 128 channels, autoencoder.py:132) with the sums for the first
    ResBlock's GroupNorm in its store loop (vqk_conv2d_thin_in_gnstats) against conv + separate statistics pass"""
    g = torch.Generator(device=DEV).manual_seed(n + h)
    x = torch.zeros(n, 8, h, w, device=DEV)
    x[:, :3] = torch.randn(n, 3, h, w, device=DEV, generator=g)
    x = x.to(BF).contiguous(memory_format=CL)
    wt = torch.zeros(128, 3, 3, 8, device=DEV)
    wt[..., :3] = torch.randn(128, 3, 3, 3, device=DEV, generator=g) / 5.0
    wq = ops.pack_weights(wt.reshape(-1), BF, 128, 8, 3, False, 0)
    gw = torch.randn(128, device=DEV, generator=g)
    gb = torch.randn(128, device=DEV, generator=g)
    y_ref = ops.raw_conv_fprop(x, wq, None, None, 3, False, 0, BF, 128, 0)
    a_ref, st_ref = ops.raw_gn_forward(y_ref, gw, gb, 32, 1e-6, True)
    y = ops.raw_conv_thin_in_gnstats(x, wq, None, 128, 32)
    assert y is not None, 'the fused kernel must serve this shape'
    a, st = ops.raw_gn_forward(y, gw, gb, 32, 1e-6, True, presummed=True)
    torch.cuda.synchronize()
    assert torch.equal(y, y_ref)
    torch.testing.assert_close(st.view(-1, 2)[:, 0], st_ref.view(-1, 2)[:, 0], rtol=0, atol=2e-5)
    torch.testing.assert_close(st.view(-1, 2)[:, 1], st_ref.view(-1, 2)[:, 1], rtol=2e-5, atol=0)
    assert float((a.float() - a_ref.float()).norm() / a_ref.float().norm()) < 1e-4
    ws = ops._gn_ws(x.device, n * 32 * 2 + n)
    assert float(ws.abs().max()) == 0.0, 'the GroupNorm workspace must be left zero'


def test_not_served_returns_none():
    x = torch.randn(1, 128, 32, 32, device=DEV).to(BF).contiguous(memory_format=CL)     # 32x32 map: single-kernel GroupNorm territory
    wq = ops.pack_weights(torch.randn(128 * 9 * 128, device=DEV) * 0.03, BF, 128, 128, 3, False, 1)
    assert ops.raw_conv_fprop_gnstats(x, wq, None, None, False, 128, 32) is None
    ws = ops._gn_ws(x.device, 128)
    torch.cuda.synchronize()
    assert float(ws.abs().max()) == 0.0


def test_unclaimed_sums_do_not_leak_into_the_next_groupnorm():
    """a producer told ``next_gn`` whose consumer turns out NOT to be a GroupNorm (the decoder's ResBlock -> Upsample conv)
    leaves sums nobody claims: the next producer / GroupNorm must clear them instead of adding to them"""
    ae = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd.modules.autoencoder')
    torch.manual_seed(3)
    c1 = ae.Conv2d(128, 128, 3, bias=True).to(DEV)
    up = ae.Upsample(128).to(DEV)
    gn = ae.GroupNorm(32, 128).to(DEV)
    x = torch.randn(4, 128, 64, 64, device=DEV).to(BF).contiguous(memory_format=CL)
    with torch.no_grad():
        saved, ops.FUSE_GN_STATS = ops.FUSE_GN_STATS, False
        ref = gn(up(c1(x)), silu=True)
        ops.FUSE_GN_STATS = True
        try:
            t = c1(x, next_gn=32)                 # sums of t are parked ... and never claimed
            assert ops.pending_gn() is not None
            got = gn(up(t, next_gn=32), silu=True)
            t2 = c1(x, next_gn=32)                # parked again; the next consumer is a GroupNorm of ANOTHER tensor
            got2 = gn(ref.to(BF), silu=False)
            ref2 = None
            ops.FUSE_GN_STATS = False
            ref2 = gn(ref.to(BF), silu=False)
        finally:
            ops.FUSE_GN_STATS = saved
    torch.cuda.synchronize()
    assert float((got.float() - ref.float()).norm() / ref.float().norm()) < 1e-3
    assert float((got2.float() - ref2.float()).norm() / ref2.float().norm()) < 1e-3
    assert float(ops._gn_ws(x.device, 4 * 64 + 4).abs().max()) == 0.0 and ops.pending_gn() is None


@pytest.mark.parametrize('cin,cout,h,ups,hr,pool', [(128, 128, 128, 0, 1, 0), (128, 256, 128, 0, 1, 1), (256, 256, 64, 0, 0, 0),
                                                    (256, 512, 128, 0, 1, 1), (128, 128, 64, 1, 0, 0)])
def test_conv_and_fused_sums_are_deterministic_and_batch_split_exact(cin, cout, h, ups, hr, pool):
    """The same conv on a batch of 4 and on its two halves, and twice on the batch: outputs AND fused GroupNorm statistics
    bit-identical (per-sample results must not depend on which block, or which launch, computes a tile -- this is what
    caught an inline-asm DOT hazard that dropped the last products of a tile in the pooled / 8-channel-group launches)."""
    hin = h >> ups
    g = torch.Generator(device=DEV).manual_seed(cin + cout + h + ups + hr + pool)
    x = (torch.randn(4, cin, hin, hin, device=DEV, generator=g) + 0.2).to(BF).contiguous(memory_format=CL)
    wt = (torch.randn(cout, 3, 3, cin, device=DEV, generator=g) / (3 * cin ** 0.5)).reshape(-1)
    res = torch.randn(4, cout, h, h, device=DEV, generator=g).to(BF).contiguous(memory_format=CL) if hr else None
    gw, gb = torch.randn(cout, device=DEV, generator=g), torch.randn(cout, device=DEV, generator=g)
    wq = ops.pack_weights(wt, BF, cout, cin, 3, False, 1)
    outs = []
    for n0, n1 in ((0, 4), (0, 2), (2, 4), (0, 4)):
        xs = x[n0:n1].contiguous(memory_format=CL)
        rs = res[n0:n1].contiguous(memory_format=CL) if hr else None
        y = ops.raw_conv_fprop_gnstats(xs, wq, None, rs, bool(ups), cout, 32, pool=bool(pool), pool_scale=0.25)
        assert y is not None
        _, st = ops.raw_gn_forward(y, gw, gb, 32, 1e-6, True, presummed=True)
        outs.append((y.clone(), st.clone()))
    torch.cuda.synchronize()
    (yf, sf), (ya, sa), (yb, sb), (yf2, sf2) = outs
    assert torch.equal(yf, yf2) and torch.equal(sf, sf2)
    assert torch.equal(yf, torch.cat([ya, yb], 0))
    assert torch.equal(sf, torch.cat([sa, sb], 0))


@pytest.mark.parametrize('n,cin,cout,h,w,ups,hb,hr,pool', [CASES[0], CASES[2], CASES[5], CASES[8], CASES[13], CASES[14]])
def test_fused_stats_in_deterministic_mode_use_one_slot_per_tile(n, cin, cout, h, w, ups, hb, hr, pool):
    """deterministic mode (round 4): the drain STORES its tile's sums in the tile's own slot and the consumer adds the slots in a
    fixed order (vqk_gn_forward_presummed_parts) -- (mean, rstd) equal to the separate pass to fp32 rounding, and two runs give
    the same bits; the upsample conv in phase form (four launches into four slot ranges) included"""
    g = torch.Generator(device=DEV).manual_seed(cin + cout + h + 3 * ups + hb + 2 * hr + 4 * pool)
    x = (torch.randn(n, cin, h, w, device=DEV, generator=g) + 0.3).to(BF).contiguous(memory_format=CL)
    wt = (torch.randn(cout, 3, 3, cin, device=DEV, generator=g) / (3 * cin ** 0.5)).reshape(-1)
    bias = torch.randn(cout, device=DEV, generator=g) if hb else None
    s = 2 if ups else 1
    res = torch.randn(n, cout, h * s, w * s, device=DEV, generator=g).to(BF).contiguous(memory_format=CL) if hr else None
    layout = ops.weight_layout(BF, n, h, w, cin, cout, 3, bool(ups))
    wq = ops.pack_weights(wt, BF, cout, cin, 3, False, layout)
    gw = torch.randn(cout, device=DEV, generator=g); gb = torch.randn(cout, device=DEV, generator=g)
    y_ref = ops.raw_conv_fprop_pooled(x, wq, bias, res, 3, bool(ups), cout, 0.25) if pool else \
        ops.raw_conv_fprop(x, wq, bias, res, 3, bool(ups), 0, BF, cout, layout)
    _, st_ref = ops.raw_gn_forward(y_ref, gw, gb, 32, 1e-6, True)
    ops.set_deterministic(True)
    try:
        runs = []
        for _ in range(2):
            y = ops.raw_conv_fprop_gnstats(x, wq, bias, res, bool(ups), cout, 32, pool=bool(pool), pool_scale=0.25)
            assert y is not None
            a, st = ops.raw_gn_forward(y, gw, gb, 32, 1e-6, True, presummed=True, conv_hw=h * s * w * s)
            runs.append((y.clone(), a.clone(), st.clone()))
        if ups and cin % 128 == 0 and res is None:                  # the phase form of the upsample conv: four slot ranges
            w4 = ops.pack_weights(wt, BF, cout, cin, 3, False, 2)
            yp = ops.raw_conv_ups_phase(x, w4, bias, cout, False, 32)
            assert yp is not None
            _, stp = ops.raw_gn_forward(yp, gw, gb, 32, 1e-6, True)   # claims the note left by the phase launches
            _, stp_ref = ops.raw_gn_stats(yp, 32, 1e-6), None
            torch.testing.assert_close(stp.view(-1, 2)[:, 0], ops.raw_gn_stats(yp, 32, 1e-6).view(-1, 2)[:, 0], rtol=0, atol=2e-5)
    finally:
        ops.set_deterministic(False)
    torch.cuda.synchronize()
    for p, q in zip(runs[0], runs[1]):
        assert torch.equal(p, q)
    assert torch.equal(runs[0][0], y_ref)
    torch.testing.assert_close(runs[0][2].view(-1, 2)[:, 0], st_ref.view(-1, 2)[:, 0], rtol=0, atol=2e-5)
    torch.testing.assert_close(runs[0][2].view(-1, 2)[:, 1], st_ref.view(-1, 2)[:, 1], rtol=2e-5, atol=0)

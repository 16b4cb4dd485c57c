"""The Entropy quantizer's distance matrix with the code tiles shared through LDS (csrc/vq.hip: vq_assign_lds_kernel + vq_lds_merge_kernel, tuning slot
VQ_LDS; reference: vqvae/modules/vector_quantizers.py:337-350 -- ``torch.cdist``-style distances, argmin, softmax statistics of
-d / T) against the register-streaming kernel it replaces: same MFMA sequence per (row, code), so the matrix and the indices are
BIT-identical; the row statistics (online log-sum-exp over another partition of the codes) to fp32 rounding."""
import importlib

import pytest
import torch

pytestmark = pytest.mark.gpu

native = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd._native')
ops = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd.ops')
DEV = 'cuda:0'


def _run(z, e, temperature, stats, slot):
    lib = native.lib()
    n, d = z.shape
    k = e.shape[0]
    f32 = dict(dtype=torch.float32, device=DEV)
    z2, e2 = torch.empty(n, **f32), torch.empty(k, **f32)
    st = ops._stream()
    native.check(lib.vqk_row_sqnorm_f32(z.data_ptr(), n, d, z2.data_ptr(), st), 'sq')
    native.check(lib.vqk_row_sqnorm_f32(e.data_ptr(), k, d, e2.data_ptr(), st), 'sq')
    idx = torch.empty(n, dtype=torch.int64, device=DEV)
    dm = torch.empty(n, k, **f32)
    lse, hrow, hsum = torch.empty(n, **f32), torch.empty(n, **f32), torch.zeros(1, **f32)
    assert lib.vqk_set_tuning(b'VQ_LDS', slot) == 0
    try:
        if stats:
            native.check(lib.vqk_vq_distances_stats_f32(z.data_ptr(), e.data_ptr(), z2.data_ptr(), e2.data_ptr(), n, k, d, 1, idx.data_ptr(),
                                                        dm.data_ptr(), temperature, lse.data_ptr(), hrow.data_ptr(), hsum.data_ptr(), st), 'dist')
        else:
            native.check(lib.vqk_vq_distances_f32(z.data_ptr(), e.data_ptr(), z2.data_ptr(), e2.data_ptr(), n, k, d, 1, idx.data_ptr(),
                                                  dm.data_ptr(), st), 'dist')
        torch.cuda.synchronize()
    finally:
        lib.vqk_reset_tuning()
        native.apply_env_tuning()
    return idx, dm, lse, hrow, hsum


@pytest.mark.parametrize('n,k', [(128, 64), (384, 320), (4096, 8192), (1024, 1024), (16384, 512)])      # K parts: 1 (few tiles), 1, 16, 4, 1 (many rows)
@pytest.mark.parametrize('stats', [False, True])
def test_lds_form_is_bit_identical_to_the_register_streaming_kernel(n, k, stats):
    g = torch.Generator(device=DEV).manual_seed(n + k)
    z = torch.randn(n, 256, device=DEV, generator=g)
    e = torch.randn(k, 256, device=DEV, generator=g) * 0.7
    e[: min(k, 40)] = z[: min(k, 40)] + 1e-3 * torch.randn(min(k, 40), 256, device=DEV, generator=g)      # near-ties and tiny distances
    e[5] = e[3]                                                                                         # an exact tie: lowest index wins
    got = _run(z, e, 0.05, stats, 1)
    want = _run(z, e, 0.05, stats, 0)
    assert torch.equal(got[0], want[0])
    assert torch.equal(got[1], want[1])
    if stats:
        # lse ~ -d_min / T is in the thousands here; h = lse - sa / s is a DIFFERENCE of two such numbers (0 for a one-hot row): both
        # kernels carry a few ulp of |lse| in it, over different partitions of the codes
        big = float(want[2].abs().max())
        assert float((got[2] - want[2]).abs().max()) <= 4e-7 * big
        assert float((got[3] - want[3]).abs().max()) <= 1e-6 * big + 1e-5
        assert abs(float(got[4]) - float(want[4])) <= (1e-6 * big + 1e-5) * n


def test_shapes_outside_the_lds_form_keep_the_old_kernel():
    g = torch.Generator(device=DEV).manual_seed(1)
    z = torch.randn(96, 256, device=DEV, generator=g)            # 96 rows: not a whole number of 128-row blocks
    e = torch.randn(104, 256, device=DEV, generator=g)           # 104 codes: not whole tiles
    a, b = _run(z, e, 0.05, True, 1), _run(z, e, 0.05, True, 0)
    for x, y in zip(a[:4], b[:4]):
        assert torch.equal(x, y)

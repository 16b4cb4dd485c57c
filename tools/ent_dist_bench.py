"""distance kernel of the entropy quantizer at (16384, 8192, 256): plain WRITE_D form against the form with the online row statistics,
and the separate row pass it replaces"""
import importlib, os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tools'))
import vqbench
native = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd._native')
lib = native.lib()
n, k, d = 16384, 8192, 256
g = torch.Generator().manual_seed(1)
z = (torch.randn(n, d, generator=g) * 0.36).cuda(); e = (torch.randn(k, d, generator=g) * 0.36).cuda()
z2 = torch.empty(n, device='cuda'); e2 = torch.empty(k, device='cuda')
s = torch.cuda.current_stream().cuda_stream
lib.vqk_row_sqnorm_f32(z.data_ptr(), n, d, z2.data_ptr(), s); lib.vqk_row_sqnorm_f32(e.data_ptr(), k, d, e2.data_ptr(), s)
idx = torch.empty(n, dtype=torch.int64, device='cuda'); dm = torch.empty(n, k, device='cuda')
lse = torch.empty(n, device='cuda'); hr = torch.empty(n, device='cuda'); hs = torch.zeros(1, device='cuda')
ps = torch.zeros(k, device='cuda'); u = torch.empty(k, device='cuda'); av = torch.zeros(1, device='cuda')
plain = lambda: lib.vqk_vq_distances_f32(z.data_ptr(), e.data_ptr(), z2.data_ptr(), e2.data_ptr(), n, k, d, 1, idx.data_ptr(), dm.data_ptr(), s)
stats = lambda: lib.vqk_vq_distances_stats_f32(z.data_ptr(), e.data_ptr(), z2.data_ptr(), e2.data_ptr(), n, k, d, 1, idx.data_ptr(), dm.data_ptr(), 0.01, lse.data_ptr(), hr.data_ptr(), hs.data_ptr(), s)
full = lambda: lib.vqk_entropy_forward_f32(dm.data_ptr(), n, k, 0.01, lse.data_ptr(), hr.data_ptr(), hs.data_ptr(), ps.data_ptr(), u.data_ptr(), av.data_ptr(), s)
pre = lambda: lib.vqk_entropy_forward_presummed_f32(dm.data_ptr(), n, k, 0.01, lse.data_ptr(), ps.data_ptr(), u.data_ptr(), av.data_ptr(), s)
for name, fn in (('distances', plain), ('distances + row statistics', stats), ('entropy_forward (rows + columns + finalize)', full), ('entropy_forward_presummed (columns + finalize)', pre)):
    print(f'{name:50s} {vqbench._time(fn, 20) * 1e6:9.1f} us')

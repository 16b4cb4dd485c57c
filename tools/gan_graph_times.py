#!/usr/bin/env python
"""Replay time of each of the VQ-GAN step's three hipGraphs (AE half, discriminator half, discriminator half + R1) and of the
two optimizer steps: where the config-4 step goes, and what the lazy R1 regularisation (every 16 steps) adds on average."""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0], '--gan', '--batch', '16', '--steps', '2', '--warmup', '2', '--no-cpu-baseline', '--no-kernel-events',
            '--no-other-configs', '--traffic', 'off', '--sustain-s', '0']
import bench


def main():
    hook = {}
    trainer_mod = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd.trainer')
    orig = trainer_mod.MiniTrainer._train_batch_gan_graphed

    def spy(self, model, batch, batch_index):
        hook['t'] = self
        return orig(self, model, batch, batch_index)
    trainer_mod.MiniTrainer._train_batch_gan_graphed = spy
    bench.main()
    tr = hook['t']
    ae_opt, disc_opt = tr.optimizers

    def t(fn, reps=10):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3
    print('AE half graph      %.2f ms' % t(tr._gan['ae'][0].replay))
    print('D half graph       %.2f ms' % t(tr._gan['d'][0].replay))
    print('D half + R1 graph  %.2f ms' % t(tr._gan['d_r1'][0].replay))
    print('AE optimizer step  %.2f ms' % t(ae_opt.step))
    print('D optimizer step   %.2f ms' % t(disc_opt.step))


if __name__ == '__main__':
    main()

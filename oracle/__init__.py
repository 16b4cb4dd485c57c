"""CPU oracle for the VQ-VAE train-step hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is product code: it may be
imported by ``tests/``, by ``__graft_entry__.smoke()`` and by ``bench.py``'s
``cpu_baseline`` leg, and only as the checker / the timed CPU baseline.  The
product package (``vqvae-vqgan-pytorch-lightning_amd``) never imports it and
fails loudly when its HIP extension is missing.
"""

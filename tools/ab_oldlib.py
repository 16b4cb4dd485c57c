#!/usr/bin/env python
"""Run a tool against an OLDER build of libvqk.so (VQK_LIB=path) that lacks entry points added since: the missing ones are
stubbed (status 0) for the duration of the tool -- same-box A/B of a kernel across rounds.  Tooling only; the product loader
(_native.lib) fails on a missing symbol.  Usage: VQK_LIB=scratch/libvqk_r4.so python tools/ab_oldlib.py tools/convbench.py [args]"""
import ctypes
import importlib
import os
import runpy
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
native = importlib.import_module('vqvae-vqgan-pytorch-lightning_amd._native')
import torch  # noqa: F401,E402  (loads the HIP runtime the library binds to)
probe = ctypes.CDLL(native.library_path(), mode=ctypes.RTLD_GLOBAL)
missing = [n for n in list(native._PROTOS) if not hasattr(probe, n)]
for n in missing:
    native._PROTOS.pop(n)
lib = native.lib()
for n in missing:
    setattr(lib, n, lambda *a, **k: 0)
print(f'[ab_oldlib] {native.library_path()}: stubbed {missing}', flush=True)
sys.argv = sys.argv[1:]
runpy.run_path(sys.argv[0], run_name='__main__')

#!/usr/bin/env python
"""a few launches of the VQ assignment kernels at (N, K, 256) for rocprofv3 --pmc passes: vq_once.py [N K iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import vqbench
a = [int(v) for v in sys.argv[1:]] + [8192, 1024, 20][len(sys.argv) - 1:]
print(vqbench.bench(a[0], a[1], iters=a[2]))

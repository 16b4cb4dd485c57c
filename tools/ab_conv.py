#!/usr/bin/env python
"""A/B of libvqk builds on ONE box (boxes differ by several %): tools/ab_conv.py [--wgrad] tagA tagB ...
runs tools/convbench.py against scratch/libvqk_<tag>.so, two interleaved repetitions, prints us per shape."""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
wgrad = '--wgrad' in args
tags = [a for a in args if not a.startswith('--')]
res = {}
for rep in range(2):
    for tag in tags:
        env = dict(os.environ, VQK_LIB=os.path.join(root, 'scratch', f'libvqk_{tag}.so'))
        env['VQK_NO_FPROP' if wgrad else 'VQK_NO_WGRAD'] = '1'
        out = subprocess.run([sys.executable, os.path.join(root, 'tools', 'convbench.py'), 'bf16', '10'], env=env,
                             capture_output=True, text=True).stdout
        for line in out.splitlines():
            m = re.match(r'\s*(\d+->\s*\d+ @\s*\d+\^2 k\d ups\d) x(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)', line)
            if m:
                us = float(m.group(6) if wgrad else m.group(4))
                res.setdefault((m.group(1), int(m.group(2))), {}).setdefault(tag, []).append(us)
print('shape'.ljust(30), *[t.rjust(9) for t in tags])
tot = {t: 0.0 for t in tags}
for (shape, cnt), d in res.items():
    best = {t: min(d[t]) for t in tags}
    for t in tags:
        tot[t] += best[t] * cnt
    print(shape.ljust(26), f'x{cnt:<2d}', *[f'{best[t]:9.1f}' for t in tags])
print('weighted total (ms)'.ljust(30), *[f'{tot[t] / 1e3:9.3f}' for t in tags])

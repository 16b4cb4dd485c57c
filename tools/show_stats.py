#!/usr/bin/env python
"""Print a rocpd_stats csv as ms/step.  usage: show_stats.py file.csv steps [top]"""
import csv, sys, re
rows = list(csv.reader(open(sys.argv[1])))[1:]
steps = float(sys.argv[2]); top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
tot = sum(int(r[2]) for r in rows)
print(f'total {tot / 1e6 / steps:.2f} ms/step over {steps:g} steps')
for r in rows[:top]:
    m = re.search(r'([A-Za-z_0-9]+)(<[^(]*>)?\(', r[0])
    name = (m.group(1) + (m.group(2) or '')) if m else r[0]
    name = re.sub(r'\(anonymous namespace\)::', '', name)[:70]
    print(f'{name:70s} {int(r[1]) / steps:7.1f}/step {int(r[2]) / 1e6 / steps:8.3f} ms/step avg {float(r[3]) / 1e3:8.1f} us')

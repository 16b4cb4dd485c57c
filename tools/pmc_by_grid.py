#!/usr/bin/env python
"""Per-(kernel, grid) average of PMC counters from a rocprofv3 rocpd database: separates the launches of one kernel by
shape.  Usage: pmc_by_grid.py results.db [substr]"""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
sub = sys.argv[2] if len(sys.argv) > 2 else ''
tables = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
t = lambda p: next(x for x in tables if x.startswith(p))
kd, ks, pe, pi = t('rocpd_kernel_dispatch'), t('rocpd_info_kernel_symbol'), t('rocpd_pmc_event'), t('rocpd_info_pmc')
cols = [r[1] for r in db.execute(f'pragma table_info({ks})')]
name_col = 'display_name' if 'display_name' in cols else 'kernel_name'
q = (f'select s.{name_col}, d.grid_size_x, d.grid_size_y, p.name, avg(e.value), count(*), avg(d.end - d.start) from {pe} e '
     f'join {pi} p on e.pmc_id = p.id join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id '
     f'group by s.{name_col}, d.grid_size_x, d.grid_size_y, p.name order by 5 desc')
for name, gx, gy, ctr, val, cnt, dur in db.execute(q):
    if sub in name:
        n = name.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:44]
        print(f'{n:44s} grid {gx:7d} x {gy:4d}  {ctr:12s} {val:14.1f}  n={cnt:3d}  avg {dur / 1e3:8.1f} us')

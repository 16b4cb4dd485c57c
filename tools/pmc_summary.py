#!/usr/bin/env python
"""Average the PMC counters of a rocprofv3 rocpd database per kernel.  Usage: pmc_summary.py results.db [substr]"""
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
q = (f'select s.{name_col}, p.name, avg(e.value), count(*) from {pe} e join {pi} p on e.pmc_id = p.id '
     f'join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id group by s.{name_col}, p.name')
res = defaultdict(dict)
for name, ctr, val, cnt in db.execute(q):
    if sub in name:
        res[name[:80]][ctr] = (val, cnt)
for name, d in res.items():
    print(name)
    for ctr, (val, cnt) in sorted(d.items()):
        print(f'   {ctr:32s} {val:18.1f}  (n={cnt})')

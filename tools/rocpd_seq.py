#!/usr/bin/env python
"""List the dispatches of kernels whose name contains a substring, in launch order, with the kernel that ran just
before each.  Usage: python tools/rocpd_seq.py results.db substring [max]"""
import sqlite3
import sys


def main(path, sub, limit=40):
    db = sqlite3.connect(path)
    tables = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = next(t for t in tables if t.startswith('rocpd_kernel_dispatch'))
    ks = next(t for t in tables if t.startswith('rocpd_info_kernel_symbol'))
    cols = [r[1] for r in db.execute(f'pragma table_info({ks})')]
    name_col = 'display_name' if 'display_name' in cols else 'kernel_name'
    dcols = [r[1] for r in db.execute(f'pragma table_info({kd})')]
    gx = 'grid_size_x' if 'grid_size_x' in dcols else ('grid_x' if 'grid_x' in dcols else None)
    rows = db.execute(f'select s.{name_col}, d.start, d.end{", d." + gx if gx else ""} from {kd} d join {ks} s on d.kernel_id = s.id '
                      f'order by d.start').fetchall()
    shown = 0
    for i, r in enumerate(rows):
        if sub in r[0]:
            prev = rows[i - 1] if i else None
            gap = (r[1] - prev[2]) if prev else 0
            print(f'{(r[2] - r[1]) / 1e3:9.1f} us  grid {r[3] if gx else "?"}  gap {gap / 1e3:7.1f} us  after {prev[0][:70] if prev else "-"}')
            shown += 1
            if shown >= int(limit):
                break


if __name__ == '__main__':
    main(*sys.argv[1:4])

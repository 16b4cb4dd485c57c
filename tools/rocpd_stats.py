#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd (sqlite) kernel trace into a --stats style table (per-kernel calls, total,
average, percentage).  Usage: python tools/rocpd_stats.py results.db [out.csv]"""
import csv
import sqlite3
import sys


def main(path, out=None):
    db = sqlite3.connect(path)
    tables = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = next(t for t in tables if t.startswith('rocpd_kernel_dispatch'))
    ks = next(t for t in tables if t.startswith('rocpd_info_kernel_symbol'))
    cols = [r[1] for r in db.execute(f'pragma table_info({ks})')]
    name_col = 'display_name' if 'display_name' in cols else 'kernel_name'
    rows = db.execute(f'select s.{name_col}, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start) '
                      f'from {kd} d join {ks} s on d.kernel_id = s.id group by s.{name_col} order by 3 desc').fetchall()
    total = sum(r[2] for r in rows)
    table = [('Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs')]
    for n, c, t, mn, mx in rows:
        table.append((n, c, t, round(t / c, 1), round(100.0 * t / total, 3), mn, mx))
    w = csv.writer(open(out, 'w', newline='') if out else sys.stdout)
    w.writerows(table)


if __name__ == '__main__':
    main(*sys.argv[1:3])

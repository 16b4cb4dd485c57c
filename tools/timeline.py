#!/usr/bin/env python
"""Timeline view of ONE train step from a rocprofv3 rocpd kernel trace: wall time, union of busy intervals,
time with two kernels in flight, idle gaps (with the kernels either side) and per-queue sums.
Usage: python tools/timeline.py results.db [step_index_from_end=2] [min_gap_us=3]"""
import os
import sqlite3
import sys


def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    return n.split('(')[0][:60]


def main(path, back=2, min_gap=3.0):
    db = sqlite3.connect(path)
    tables = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = next(t for t in tables if t.startswith('rocpd_kernel_dispatch'))
    ks = next(t for t in tables if t.startswith('rocpd_info_kernel_symbol'))
    cols = [r[1] for r in db.execute(f'pragma table_info({ks})')]
    name_col = 'display_name' if 'display_name' in cols else 'kernel_name'
    rows = db.execute(f'select s.{name_col}, d.start, d.end, d.queue_id from {kd} d join {ks} s on d.kernel_id = s.id '
                      f'order by d.start').fetchall()
    # a step = from one adamw launch to the next
    marks = [i for i, r in enumerate(rows) if 'adamw' in r[0]]
    if len(marks) < back + 1:
        raise SystemExit('not enough steps in the trace')
    lo, hi = marks[-back - 1] + 1, marks[-back] + 1
    step = rows[lo:hi]
    t0, t1 = rows[marks[-back - 1]][2], step[-1][2]
    print(f'step: {len(step)} kernels, wall {(t1 - t0) / 1e3:.1f} us (end of previous AdamW -> end of this AdamW)')
    ev = []
    for n, s, e, q in step:
        ev.append((s, 1)); ev.append((e, -1))
    ev.sort()
    busy = two = 0
    depth, last = 0, t0
    for t, d in ev:
        if depth >= 1: busy += t - last
        if depth >= 2: two += t - last
        depth += d; last = t
    print(f'busy (>=1 kernel) {busy / 1e3:.1f} us, >=2 kernels {two / 1e3:.1f} us, idle {(t1 - t0 - busy) / 1e3:.1f} us, '
          f'sum of kernel durations {sum(e - s for _, s, e, _ in step) / 1e3:.1f} us')
    qs = {}
    for n, s, e, q in step:
        qs.setdefault(q, [0, 0]); qs[q][0] += 1; qs[q][1] += e - s
    for q, (c, t) in qs.items():
        print(f'  queue {q}: {c} kernels, {t / 1e3:.1f} us')
    mark = os.environ.get('TL_MARK')                            # kernels to place on the step's time axis (e.g. TL_MARK=probe)
    if mark:
        for n, s, e, q in step:
            if mark in n:
                print(f'  mark q{q} {short(n):40s} start {(s - t0) / 1e3:9.1f} us  end {(e - t0) / 1e3:9.1f} us  ({(e - s) / 1e3:.1f} us)')
    # idle gaps
    gaps = []
    depth, idle_start, prev = 0, t0, 'AdamW(prev)'
    order = sorted([(s, 1, n) for n, s, e, q in step] + [(e, -1, n) for n, s, e, q in step], key=lambda x: (x[0], -x[1]))
    for t, d, n in order:
        if depth == 0 and d == 1 and t - idle_start > min_gap * 1e3:
            gaps.append(((t - idle_start) / 1e3, prev, n))
        depth += d
        if depth == 0:
            idle_start, prev = t, n
    tot = sum(g[0] for g in gaps)
    print(f'{len(gaps)} idle gaps > {min_gap} us, {tot:.1f} us in total; largest:')
    for g, a, b in sorted(gaps, reverse=True)[:25]:
        print(f'  {g:8.1f} us  after {short(a)}  before {short(b)}')
    # context of the largest gap: the kernels either side of it
    if gaps:
        g, a, b = max(gaps)
        names = [short(n) for n, s_, e, q in step]
        starts = [s_ for n, s_, e, q in step]
        order2 = sorted(range(len(step)), key=lambda i: starts[i])
        # index of the first kernel starting after the gap
        ends = [e for n, s_, e, q in step]
        for pos, i in enumerate(order2):
            if pos and starts[i] - max(ends[j] for j in order2[:pos]) > (g - 0.5) * 1e3:
                lo_, hi_ = max(0, pos - 8), min(len(order2), pos + 8)
                print(f'around the largest gap ({g:.1f} us):')
                for k in range(lo_, hi_):
                    j = order2[k]
                    print(f'   {"-->" if k == pos else "   "} q{step[j][3]} {names[j]:60s} {(ends[j] - starts[j]) / 1e3:8.1f} us')
                break
    # time the busiest queue (the main stream) sits idle while another queue runs: joins waiting on side-stream kernels
    main_q = max(qs, key=lambda q: qs[q][1])
    main_iv = sorted((s_, e) for n, s_, e, q in step if q == main_q)
    side_iv = sorted((s_, e, n) for n, s_, e, q in step if q != main_q)
    waits = {}
    tot_wait = 0.0
    for (s0, e0), (s1, e1) in zip(main_iv, main_iv[1:]):
        if s1 - e0 <= 1500:
            continue
        for ss, se, n in side_iv:                              # side kernels overlapping the main-queue hole [e0, s1]
            lo_, hi_ = max(ss, e0), min(se, s1)
            if hi_ > lo_:
                waits[short(n)] = waits.get(short(n), 0.0) + (hi_ - lo_) / 1e3
                tot_wait += (hi_ - lo_) / 1e3
    print(f'main queue {main_q} idle while another queue runs: {tot_wait:.1f} us in total')
    for n, t in sorted(waits.items(), key=lambda kv: -kv[1])[:6]:
        print(f'  waiting on {n:56s} {t:9.1f} us')
    # per-kernel sums in this step
    agg = {}
    for n, s, e, q in step:
        a = agg.setdefault(short(n), [0, 0]); a[0] += 1; a[1] += e - s
    print('per kernel in this step:')
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
        print(f'  {n:60s} {c:4d} {t / 1e3:9.1f} us')


if __name__ == '__main__':
    a = sys.argv
    main(a[1], int(a[2]) if len(a) > 2 else 2, float(a[3]) if len(a) > 3 else 3.0)

#!/usr/bin/env python
"""Overlap statistics of a rocprofv3 kernel trace (rocpd sqlite): over the steady-state window [--skip fraction of the densest segment,
end], the wall time, the time with 0 / 1 / 2 / >= 3 kernels in flight, and per kernel family the mean number of OTHER kernels in
flight while it runs.  python tools/timeline_overlap.py x_results.db [--skip 0.5]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    skip = float(sys.argv[sys.argv.index('--skip') + 1]) if '--skip' in sys.argv else 0.5
    c = sqlite3.connect(db)
    rows = c.execute('select name, start, end from kernels order by start').fetchall()
    # the densest segment of the trace (kernels less than 2 ms apart): the timed loop; its first `skip` fraction is dropped
    segs, cur, hi = [], [rows[0]], rows[0][2]
    for r in rows[1:]:
        if r[1] - hi > 2e6:
            segs.append(cur)
            cur = []
        cur.append(r)
        hi = max(hi, r[2])
    segs.append(cur)
    rows = max(segs, key=len)
    t0, t1 = rows[0][1], max(r[2] for r in rows)
    lo = t0 + (t1 - t0) * skip
    rows = [(n.split('(')[0].replace('void ', '').replace('lp::', ''), s, e) for n, s, e in rows if s >= lo]
    ev = []
    for i, (n, s, e) in enumerate(rows):
        ev.append((s, 1, i))
        ev.append((e, -1, i))
    ev.sort()
    depth, last = 0, ev[0][0]
    hist = {}
    alone = {}
    active = set()
    for t, d, i in ev:
        dt = t - last
        hist[min(depth, 3)] = hist.get(min(depth, 3), 0) + dt
        for a in active:
            k = alone.setdefault(rows[a][0], [0, 0])
            k[0] += dt * (depth - 1)
            k[1] += dt
        last = t
        depth += d
        if d > 0:
            active.add(i)
        else:
            active.discard(i)
    wall = ev[-1][0] - ev[0][0]
    print('window %.3f ms, %d kernels' % (wall / 1e6, len(rows)))
    for k in sorted(hist):
        print('  %s kernels in flight: %6.2f %%' % ('>=3' if k == 3 else str(k), 100.0 * hist[k] / wall))
    print('%-44s %10s %8s' % ('kernel', 'busy ms', 'others'))
    for n, (w, b) in sorted(alone.items(), key=lambda kv: -kv[1][1])[:24]:
        print('%-44s %10.3f %8.2f' % (n[:44], b / 1e6, w / max(b, 1)))


if __name__ == '__main__':
    main()

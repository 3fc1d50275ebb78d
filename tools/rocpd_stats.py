#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace into the text table committed under
profiles/:  python tools/rocpd_stats.py gpurun_out/prof/x_results.db [--skip-first K]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    c = sqlite3.connect(db)
    rows = c.execute('select name, duration, grid_x, grid_y, workgroup_x, lds_size, vgpr_count, '
                     'accum_vgpr_count, sgpr_count from kernels order by start').fetchall()
    agg = {}
    for name, dur, gx, gy, wx, lds, vg, ag, sg in rows:
        name = name.split('(')[0]
        a = agg.setdefault(name, [0, 0.0, 1e30, 0.0, vg, ag, sg, lds])
        a[0] += 1
        a[1] += dur
        a[2] = min(a[2], dur)
        a[3] = max(a[3], dur)
    tot = sum(a[1] for a in agg.values())
    print('%-64s %7s %12s %10s %10s %10s %6s %5s %5s %5s %7s' %
          ('kernel', 'calls', 'total_us', 'avg_us', 'min_us', 'max_us', '%', 'vgpr', 'agpr', 'sgpr', 'lds'))
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print('%-64s %7d %12.1f %10.2f %10.2f %10.2f %6.2f %5d %5d %5d %7d' %
              (name[:64], a[0], a[1] / 1e3, a[1] / a[0] / 1e3, a[2] / 1e3, a[3] / 1e3, 100 * a[1] / tot,
               a[4], a[5], a[6], a[7]))
    print('total kernel time %.3f ms over %d dispatches' % (tot / 1e6, len(rows)))


if __name__ == '__main__':
    main()

#!/usr/bin/env python
"""Aggregate rocprofv3 PMC counters (rocpd sqlite) per kernel name:
    python tools/rocpd_pmc.py gpurun_out/pmc1/p1_results.db [more.db ...]"""
import sqlite3
import sys

for db in sys.argv[1:]:
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info('pmc_events')")]
    # views differ slightly between versions: discover the useful columns
    name_col = 'counter_name' if 'counter_name' in cols else 'pmc_name' if 'pmc_name' in cols else None
    q = None
    try:
        q = c.execute('select name, counter_name, sum(counter_value), count(*) from pmc_events '
                      'group by name, counter_name').fetchall()
    except Exception as e:            # fall back: print the schema so the query can be fixed
        print(db, 'query failed:', e, cols)
        continue
    agg = {}
    for kname, cname, v, n in q:
        agg.setdefault(kname.split('(')[0], {})[cname] = (v, n)
    names = sorted({cn for d in agg.values() for cn in d})
    print('#', db)
    print('%-52s' % 'kernel' + ''.join('%22s' % n for n in names) + '%8s' % 'calls')
    for k, d in sorted(agg.items(), key=lambda kv: -max(v[0] for v in kv[1].values())):
        print('%-52s' % k[:52] + ''.join('%22.4g' % d.get(n, (0, 0))[0] for n in names) +
              '%8d' % max(v[1] for v in d.values()))

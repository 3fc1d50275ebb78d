#!/usr/bin/env python
"""FETCH_SIZE / WRITE_SIZE rocpd databases (separate --pmc passes over tools/profile_ops.py
--reps R) -> profiles/r01_traffic.json: HBM bytes per forward for every kernel.
  python tools/pmc_traffic.py fetch.db write.db FORWARDS [out.json] [commit] [note] [config-json]
config-json = {"arch": .., "size": .., "batch": .., "storage": ..} of the profiled run: bench.py quotes a file only for
the configuration it was measured on."""
import json
import sqlite3
import sys


def total(db, counter):
    c = sqlite3.connect(db)
    out = {}
    for name, v in c.execute('select name, sum(counter_value) from pmc_events where counter_name=? group by name',
                             (counter,)):
        k = name.split('(')[0].replace('void ', '').replace('lp::', '')
        # families as lp::last_kernel_tag names them: depthwise kernels keep their <K[,S]>, the rest no template
        k = k.split('<')[0] if not k.startswith('dw') else k.replace(', true>', '>').replace(', false>', '>').replace(', ', ',')
        out[k] = out.get(k, 0.0) + v
    return out


fetch_db, write_db, fwd = sys.argv[1], sys.argv[2], int(sys.argv[3])
out_path = sys.argv[4] if len(sys.argv) > 4 else 'profiles/r02_traffic.json'
commit = sys.argv[5] if len(sys.argv) > 5 else 'unknown' 
f, w = total(fetch_db, 'FETCH_SIZE'), total(write_db, 'WRITE_SIZE')
note = sys.argv[6] if len(sys.argv) > 6 else 'per forward of 64 images + 64 mirrored, XS@256'
config = json.loads(sys.argv[7]) if len(sys.argv) > 7 else {'arch': 'search-XS', 'size': 256, 'batch': 64, 'storage': 'f32'}
res = {'config': config, 'note': 'KiB counters * 1024; FETCH_SIZE doubled (gfx950 counts 64 B per 128-B request, '
               'MI355X_MICROARCH.md HBM); ' + note,
       'forwards_profiled': fwd, 'commit': commit, 'kernels': {}}
for k in sorted(set(f) | set(w)):
    rd = 2.0 * f.get(k, 0.0) * 1024 / fwd
    wr = w.get(k, 0.0) * 1024 / fwd
    res['kernels'][k] = {'read_bytes_per_forward': int(rd), 'write_bytes_per_forward': int(wr),
                         'hbm_bytes_per_forward': int(rd + wr)}
json.dump(res, open(out_path, 'w'), indent=1)
print(json.dumps(res, indent=1))

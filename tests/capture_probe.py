#!/usr/bin/env python
"""Probe (run by tests/test_gpu_real_shapes.py in a SUBPROCESS, because a broken capture can abort the interpreter):
what happens to PoseEngine's hipGraph captures while a second host thread polls events / streams -- the access
pattern of torch.distributed's RCCL watchdog thread -- under both capture error modes.

    python tests/capture_probe.py thread_local|global      -> one JSON line

Reports: records right (always required), whether graphs were captured or the engine fell back to eager launches,
how many of the poller's calls raised, and that the eager fallback works right after a failed capture."""
import json
import os
import sys
import threading
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from litepose_amd import arch_zoo, config, engine  # noqa: E402
from oracle import inference_ref, synth  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else 'thread_local'
arch = arch_zoo.get('search-XS')
cfg = config.apply_arch(config.get_cfg(), arch)
sd = synth.make_state_dict(arch, seed=1234, head_gain=0.25)
N, R = 8, 128
x = synth.make_images(N, R, seed=910).cuda()
off0, off1 = synth.lowres_offsets(911, N, 14, R)
f0, f1 = synth.flip_offsets(off0, off1, inference_ref.FLIP_CONFIG['CROWDPOSE'])
offs = (torch.from_numpy(np.concatenate([off0, f0])).cuda(), torch.from_numpy(np.concatenate([off1, f1])).cuda())
eng0 = engine.PoseEngine(cfg, arch, sd, person_capacity=30)
ra, rc, rs = [t.clone() for t in eng0.infer_batch(x, offsets=offs)]
torch.cuda.synchronize()

stop = threading.Event()
polls = {'ok': 0, 'raised': 0, 'last': ''}


def poller():
    torch.cuda.set_device(0)
    st = torch.cuda.Stream()
    ev = torch.cuda.Event()
    ev.record(st)
    while not stop.is_set():
        try:
            ev.query()
            st.query()
            polls['ok'] += 1
        except Exception as e:                   # what the watchdog would see while a capture is open
            polls['raised'] += 1
            polls['last'] = str(e).splitlines()[0][:100]


th = threading.Thread(target=poller, daemon=True)
eng = engine.PoseEngine(cfg, arch, sd, person_capacity=30, capture_mode=mode)
th.start()
ok = True
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    eng.prepare(x, offsets=offs)
    for it in range(12):
        with eng.submit(x, offsets=offs) as (a, c, s):
            good = torch.equal(c, rc)
            for n in range(N):
                k = min(int(c[n]), 30)
                good = good and torch.equal(a[n, :k], ra[n, :k]) and torch.equal(s[n, :k], rs[n, :k])
            ok = ok and bool(good)
        torch.cuda.synchronize()
stop.set()
th.join(timeout=10)
out = dict(eng.graph_stats(), mode=mode, records_ok=ok, polls_ok=polls['ok'], polls_raised=polls['raised'],
           poll_error=polls['last'])
print(json.dumps(out))
sys.stdout.flush()
os._exit(0)          # skip interpreter teardown: a graph object of a broken capture may throw in its destructor

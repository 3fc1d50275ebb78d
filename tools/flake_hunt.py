#!/usr/bin/env python
"""Hunt for the rare wrong batch of the pipelined serving schedule (PoseEngine.submit: two NET streams + one AE
stream, four buffer sets, hipGraph replay) -- tests/test_gpu_real_shapes.py::
test_submit_split_schedule_stress_two_inputs_in_flight fails about once in a few dozen runs with ONE batch of 48 off.

The test's loop, many more iterations, and on every mismatch a post-mortem of the buffer set that produced it:
which record fields differ, and whether the set's merged maps (`mid`, `det` -- what the NET stage left for the AE
stage) equal the maps of a clean run of the same input.  NET maps equal + records differ -> the AE stage raced;
NET maps differ -> a network kernel or the staging refill did.

    python tools/flake_hunt.py --iters 20000 [--batch 8] [--size 256] [--eager]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from litepose_amd import arch_zoo, config, engine  # noqa: E402
from oracle import inference_ref, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--iters', type=int, default=20000)
ap.add_argument('--batch', type=int, default=8)
ap.add_argument('--size', type=int, default=256)
ap.add_argument('--arch', default='search-XS')
ap.add_argument('--storage', default='f32', choices=['f32', 'bf16'])
ap.add_argument('--eager', action='store_true', help='LP_GRAPH=0: eager launches instead of graph replay')
ap.add_argument('--max-report', type=int, default=12)
ap.add_argument('--opt', action='append', default=[],
                help='kernel-family switch key=value of the HUNTED engine (lp_net_set_option), repeatable; the clean '
                     'reference engine keeps the defaults')
ap.add_argument('--diag', action='store_true',
                help='DESIGN 5b diagnostics: --opt stem=0 --opt diag_dwpw=1 and a dump of the self-checking bias fetch log '
                     '(lp_diag_read) at the end')
a = ap.parse_args()
if a.eager:
    os.environ['LP_GRAPH'] = '0'
arch = arch_zoo.get(a.arch)
cfg = config.apply_arch(config.get_cfg(), arch)
sd = synth.make_state_dict(arch, seed=1234, head_gain=0.25)
N, R = a.batch, a.size


def offsets(seed):
    off0, off1 = synth.lowres_offsets(seed, N, 14, R)
    f0, f1 = synth.flip_offsets(off0, off1, inference_ref.FLIP_CONFIG['CROWDPOSE'])
    return (torch.from_numpy(np.concatenate([off0, f0])).cuda(), torch.from_numpy(np.concatenate([off1, f1])).cuda())


xs = [synth.make_images(N, R, seed=700 + k).cuda() for k in range(2)]
offs_all = [offsets(800 + k) for k in range(2)]
eng = engine.PoseEngine(cfg, arch, sd, person_capacity=30, storage=a.storage, options=engine.options_from_env())
if a.diag:
    a.opt = ['stem=0', 'diag_dwpw=1'] + a.opt
for kv in a.opt:
    k_, v_ = kv.split('=')
    eng.model.set_option(k_, int(v_))
if a.opt:
    print('options of the hunted engine:', ' '.join(a.opt))
# clean references: un-pipelined, single stream, one input at a time (+ the maps the NET stage leaves behind)
ref = []
probe = engine.PoseEngine(cfg, arch, sd, person_capacity=30, pipeline_halves=False, storage=a.storage,
                          options=engine.options_from_env())
for k in range(2):
    r = []
    for rep in range(3):
        ans, cnt, sc = probe.infer_batch(xs[k], offsets=offs_all[k])
        torch.cuda.synchronize()
        b = probe._buffers(N, R, R)
        # `det` exists on the 'dm' / 'maps' AE paths only; the default since round 5 ('mid') never materialises it
        r.append((ans.clone(), cnt.clone(), sc.clone(), b['tta_ws'].clone(),
                  b['det'].clone() if b['det'] is not None else torch.zeros(1, device='cuda'), b['net_ws'].clone()))
    for rep in (1, 2):
        assert all(torch.equal(r[0][i], r[rep][i]) for i in range(5)), 'the clean reference is not reproducible'
    ref.append(r[0])
print('reference: %d + %d persons' % (int(ref[0][1].sum()), int(ref[1][1].sum())))

# block-boundary taps and where they live in a workspace of this shape (lp_net_tap_offset)
import ctypes as C  # noqa: E402
from litepose_amd import _native as nv  # noqa: E402
TAPS = ['first'] + ['stage.%d.%d' % (s_, b_) for s_, nb in enumerate((6, 8, 10, 10)) for b_ in range(nb)] + \
       ['deconv.0', 'deconv.1', 'deconv.2']
tap_off = {}
for name in TAPS:
    if a.storage != 'f32':
        break                                   # per-tap post-mortem: fp32 workspaces only
    c_ = C.c_int64(0)
    off = nv.lib().lp_net_tap_offset(probe.model._h, name.encode(), 2 * N, R, R, C.byref(c_))
    if off >= 0:
        tap_off[name] = (int(off), int(c_.value))
TAPS = [t for t in TAPS if t in tap_off]
# taps that are reproducible between clean runs (every tensor the forward writes is; unused buffers are not read)
for name in TAPS:
    off, cnt_f = tap_off[name]
    assert torch.equal(r[0][5][off:off + 4 * cnt_f], r[1][5][off:off + 4 * cnt_f]), name

depth = eng.pipeline_depth()
nset = eng.buffer_sets()
stage = [(xs[0].clone(), tuple(o.clone() for o in offs_all[0])) for _ in range(nset)]
pend, bad = [], []
lane_of = {}


def collect():
    it, k, h = pend.pop(0)
    a_, c_, s_ = h.result()
    ok = torch.equal(c_, ref[k][1]) and torch.equal(a_, ref[k][0]) and torch.equal(s_, ref[k][2])
    if not ok:
        torch.cuda.synchronize()
        ln = eng._lanes[it % nset]
        b = ln['eng']._buffers(N, R, R)
        mid_ok = bool(torch.equal(b['tta_ws'], ref[k][3]))
        det_ok = b['det'] is None or bool(torch.equal(b['det'], ref[k][4]))
        # the other input's maps?  (a set that ran on a stale staging buffer)
        mid_other = bool(torch.equal(b['tta_ws'], ref[1 - k][3]))
        dc = (c_ != ref[k][1]).nonzero().flatten().tolist()
        da = (a_ != ref[k][0]).nonzero()
        ds = (s_ != ref[k][2]).nonzero()
        rec_other = bool(torch.equal(c_, ref[1 - k][1]) and torch.equal(a_, ref[1 - k][0]))
        info = {'it': it, 'set': it % nset, 'input': k, 'mid_equal': mid_ok, 'det_equal': det_ok,
                'mid_is_other_input': mid_other, 'records_are_other_input': rec_other,
                'count_diff_images': dc[:8], 'kpts_diff_elems': int(da.shape[0]),
                'kpts_diff_images': sorted(set(da[:, 0].tolist()))[:8] if da.shape[0] else [],
                'scores_diff_elems': int(ds.shape[0])}
        if not mid_ok:
            m = b['tta_ws'].view(torch.float32)[:ref[k][3].numel() // 4]
            r_ = ref[k][3].view(torch.float32)
            d = (m != r_).nonzero().flatten()
            info['mid_diff_elems'] = int(d.numel())
            if d.numel():
                per_img = m.numel() // N
                info['mid_diff_images'] = sorted(set((d // per_img).tolist()))[:8]
                info['mid_max_abs'] = float((m - r_).abs().max())
        if not det_ok:
            # where and how the projected heatmaps differ (the det-only projection is the only writer of this tensor): pixel
            # runs, the values found, the clean values, the OTHER input's clean values there, zeros
            dd = b['det'].flatten()
            d0 = ref[k][4].flatten()
            d1 = ref[1 - k][4].flatten()
            ix = (dd != d0).nonzero().flatten()
            info['det_diff_elems'] = int(ix.numel())
            J_, H_, W_ = b['det'].shape[1:]
            sel = ix[:48]
            got, want, other = dd[sel].tolist(), d0[sel].tolist(), d1[sel].tolist()
            ys, xs_ = (ix // W_) % H_, ix % W_
            info['det_diff_parity_yx'] = {'%d%d' % (py, px): int(((ys % 2 == py) & (xs_ % 2 == px)).sum()) for py in (0, 1) for px in (0, 1)}
            info['det_diff_rows'] = sorted(set(ys.tolist()))[:12]
            info['det_diff_cells_mod32'] = sorted(set(((xs_ // 2) % 32).tolist()))
            info['det_diff_equal_other_input'] = int((dd[ix] == d1[ix]).sum())
            info['det_diff_zero'] = int((dd[ix] == 0).sum())
            info['det_diff_first'] = [((int(i) // (J_ * H_ * W_)), (int(i) // (H_ * W_)) % J_, (int(i) // W_) % H_, int(i) % W_,
                                       float('%.6g' % g), float('%.6g' % w), float('%.6g' % o))
                                      for i, g, w, o in zip(sel.tolist(), got, want, other)]
        # which block boundary of the set's own network workspace differs first (one buffer per tensor: every tap
        # of the forward that produced this batch is still there)
        ws, ws_ref = b['net_ws'], ref[k][5]
        first = None
        for name in TAPS:
            off, cnt_f = tap_off[name]
            x1 = ws[off:off + 4 * cnt_f].view(torch.float32)
            x0 = ws_ref[off:off + 4 * cnt_f].view(torch.float32)
            if not torch.equal(x1, x0):
                dd = (x1 != x0).nonzero().flatten()
                per = cnt_f // (2 * N)
                first = {'tap': name, 'elems': int(dd.numel()), 'images': sorted(set((dd // per).tolist()))[:6],
                         'max_abs': float((x1 - x0).abs().max()),
                         'first_idx_in_image': int(dd[0] % per), 'plane_elems': per}
                break
        info['first_bad_tap'] = first
        bad.append(info)
        if len(bad) <= a.max_report:
            print('MISMATCH', info)
            sys.stdout.flush()
    h.release()


for it in range(a.iters):
    k = (it // 3 + it) % 2
    xb, ob = stage[it % nset]
    xb.copy_(xs[k])
    for dst, src in zip(ob, offs_all[k]):
        dst.copy_(src)
    pend.append((it, k, eng.submit(xb, offsets=ob)))
    if len(pend) > depth:
        collect()
while pend:
    collect()
torch.cuda.synchronize()
print('iterations %d, mismatching batches %d (%.3g per batch); graphs: %s' % (a.iters, len(bad), len(bad) / a.iters,
                                                                        eng.graph_stats()))

if a.diag:
    import ctypes as C
    from litepose_amd import _native as nv
    buf = (C.c_uint32 * (1 + 16 * 256))()
    n_ev = nv.lib().lp_diag_read(C.cast(buf, C.c_void_p), len(buf), 1)
    print('diag_dwpw: %d times the vector-loaded bias registers disagreed with the scalar-cache copy (after the load / before the use)' % n_ev)
    for e in range(min(max(n_ev, 0), 256)):
        r = buf[1 + 16 * e: 1 + 16 * (e + 1)]
        bm, bm2 = r[3] | (r[4] << 32), r[5] | (r[6] << 32)
        hw = r[9]
        # HW_ID (gfx9): wave_id [3:0], simd_id [5:4], pipe_id [7:6], cu_id [11:8], sh_id [12], se_id [15:13] (gfx950: [16:13])
        print('  %s wg %6d wave %d dword %2d  bad lanes %016x (%2d)  still bad on re-fetch %016x  got %08x want %08x  '
              'HW_ID %08x (wave slot %d simd %d cu %d se %d) xcc %d  t %d  K %d grid %d'
              % ('epilogue' if r[2] >> 8 else 'load    ', r[0], r[1], r[2] & 255, bm, bin(bm).count('1'), bm2, r[7], r[8], hw, hw & 15, (hw >> 4) & 3, (hw >> 8) & 15,
                 (hw >> 13) & 15, r[10] & 15, r[11] | (r[12] << 32), r[13], r[14]))

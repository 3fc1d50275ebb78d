"""fp32 torch-functional restatement of ``LitePose.forward``.  TEST INFRASTRUCTURE.

The conv net is floating-point work, so the oracle for it is a plain torch fp32
reference of the same ops in the reference's op order (conv -> BN(eval) -> act,
unfused), driven by a reference-format ``state_dict``:
  * stem         /root/reference/lib/models/pose_mobilenet.py:36-41, layers.py:18-24
  * InvBottleneck /root/reference/lib/models/layers/layers.py:90-118
  * deconv head  /root/reference/lib/models/pose_mobilenet.py:143-156 (ConvT k4 s2 p1)
  * SepConv2d    /root/reference/lib/models/layers/layers.py:120-133
Pinned by tests/golden/gen_golden.py against the real module (bit-identical at
equal thread count).
"""
import torch
import torch.nn.functional as F

from . import spec

BN_EPS = 1e-5


def _bn(x, sd, p):
    return F.batch_norm(x, sd[p + '.running_mean'], sd[p + '.running_var'],
                        sd[p + '.weight'], sd[p + '.bias'], False, 0.0, BN_EPS)


def _relu6(x):
    return torch.clamp(x, 0.0, 6.0)


def stem(x, sd):
    x = _relu6(_bn(F.conv2d(x, sd['first.0.0.weight'], None, 2, 1), sd, 'first.0.1'))
    x = _relu6(_bn(F.conv2d(x, sd['first.1.0.weight'], None, 1, 1, 1, 32), sd, 'first.1.1'))
    x = _bn(F.conv2d(x, sd['first.2.weight']), sd, 'first.3')
    return x


def inv_bottleneck(x, sd, p, blk, taps=None):
    k = blk['k']
    out = _relu6(_bn(F.conv2d(x, sd[p + '.inv.0.weight']), sd, p + '.inv.1'))
    if taps is not None:
        taps[p + '.inv'] = out
    out = _relu6(_bn(F.conv2d(out, sd[p + '.depth_conv.0.weight'], None, blk['stride'],
                              k // 2, 1, blk['feat']), sd, p + '.depth_conv.1'))
    if taps is not None:
        taps[p + '.depth_conv'] = out
    out = _bn(F.conv2d(out, sd[p + '.point_conv.0.weight']), sd, p + '.point_conv.1')
    if blk['residual']:
        out = out + x
    return out


def sep_conv(x, sd, p):
    c = x.shape[1]
    out = F.relu(_bn(F.conv2d(x, sd[p + '.0.weight'], None, 1, 2, 1, c), sd, p + '.1'))
    return F.conv2d(out, sd[p + '.3.weight'])


def forward(x, sd, arch, head=None, taps=None):
    """Returns [out0 (N, oup0, H/4, W/4), out1 (N, oup1, H/2, W/2)].

    ``taps``: optional dict that receives every block-boundary tensor."""
    d = spec.derive(arch, head)
    x = stem(x, sd)
    x_list = [x]
    if taps is not None:
        taps['first'] = x
    for s, blocks in enumerate(d['stages']):
        for b, blk in enumerate(blocks):
            x = inv_bottleneck(x, sd, 'stage.%d.%d' % (s, b), blk, taps)
            if taps is not None:
                taps['stage.%d.%d' % (s, b)] = x
        x_list.append(x)
    outs = []
    refined = x_list[-1]
    raw = x_list[-2]
    for i in range(len(d['deconv'])):
        r = F.conv_transpose2d(refined, sd['deconv_refined.%d.weight' % i], None, 2, 1)
        w = F.conv_transpose2d(raw, sd['deconv_raw.%d.weight' % i], None, 2, 1)
        refined = F.relu(_bn(r + w, sd, 'deconv_bnrelu.%d.0' % i))
        if taps is not None:
            taps['deconv.%d' % i] = refined
        raw = x_list[-i - 3]
        if i > 0:
            outs.append(sep_conv(refined, sd, 'final_refined.%d.conv' % (i - 1)) +
                        sep_conv(raw, sd, 'final_raw.%d.conv' % (i - 1)))
    return outs


# ---------------------------------------------------------------------------------------
# bf16-storage emulation (SURVEY.md 8 row g; the reference's reduced-precision eval path is
# valid.py:152-153 -> lib/fp16_utils/fp16util.py:87-91 network_to_half).  The device path
# keeps ACTIVATIONS and (BN-folded) WEIGHTS in bf16 and accumulates in fp32; this restates
# exactly that in torch fp32: fold BN in float64 -> fp32 -> round-to-nearest-even to bf16,
# fp32 convolutions, fp32 bias / activation / residual, then ONE bf16 rounding where the
# device stores the tensor.  The two head 1x1s stay fp32 on output (the AE stage reads fp32).
# Not bit-identical to the device (fp32 summation order differs and may flip a bf16
# rounding); tests bound the difference in bf16 ulps and report it against `forward`.
# ---------------------------------------------------------------------------------------
def _rb(x):
    return x.to(torch.bfloat16).to(torch.float32)


def _fold(sd, wkey, bn, transposed=False):
    w = sd[wkey].double()
    if bn is None:
        return _rb(w.float()), None
    s = sd[bn + '.weight'].double() / torch.sqrt(sd[bn + '.running_var'].double() + BN_EPS)
    sh = sd[bn + '.bias'].double() - sd[bn + '.running_mean'].double() * s
    if transposed:                      # ConvTranspose2d weight is [Cin][Cout][k][k]
        w = w * s.view(1, -1, 1, 1)
    else:
        w = w * s.view(-1, 1, 1, 1)
    return _rb(w.float()), sh.float()


def bf16_plan(sd, arch, head=None):
    """The bf16-storage network as a list of ops ``(name, inputs, fn)`` in launch order: ``fn(*tensors)``
    maps the named inputs ('x' = the fp32 image, otherwise the ``name`` of an earlier op) to the op's output
    AFTER the rounding the device applies where it stores the tensor.  Names are the device plan's op names
    (lp_net_tap accepts them), so a test can feed every op the DEVICE's own inputs and compare one layer at
    a time -- end to end the 40-layer residual trunk amplifies single bf16 rounding flips chaotically."""
    d = spec.derive(arch, head)
    ops = []

    def conv(name, src, wkey, bn, stride=1, pad=0, groups=1, act=None, res=None):
        w, b = _fold(sd, wkey, bn)

        def fn(x, r=None):
            y = F.conv2d(x, w, b, stride, pad, 1, groups)
            if act == 'relu6':
                y = torch.clamp(y, 0.0, 6.0)
            elif act == 'relu':
                y = F.relu(y)
            if r is not None:
                y = y + r
            return _rb(y)
        ops.append((name, [src] + ([res] if res is not None else []), fn))
        return name

    x = conv('stem.conv3x3s2', 'x', 'first.0.0.weight', 'first.0.1', 2, 1, 1, 'relu6')
    x = conv('stem.dw3', x, 'first.1.0.weight', 'first.1.1', 1, 1, 32, 'relu6')
    x = conv('stem.pw', x, 'first.2.weight', 'first.3')
    x_list = [x]
    for s, blocks in enumerate(d['stages']):
        for b, blk in enumerate(blocks):
            p = 'stage.%d.%d' % (s, b)
            e = conv(p + '.inv', x, p + '.inv.0.weight', p + '.inv.1', act='relu6')
            e = conv(p + '.depth_conv', e, p + '.depth_conv.0.weight', p + '.depth_conv.1', blk['stride'],
                     blk['k'] // 2, blk['feat'], 'relu6')
            x = conv(p + '.point_conv', e, p + '.point_conv.0.weight', p + '.point_conv.1',
                     res=x if blk['residual'] else None)
        x_list.append(x)
    refined, raw = x_list[-1], x_list[-2]
    for i in range(len(d['deconv'])):
        bn = 'deconv_bnrelu.%d.0' % i
        wr, sh = _fold(sd, 'deconv_refined.%d.weight' % i, bn, transposed=True)
        ww, _ = _fold(sd, 'deconv_raw.%d.weight' % i, bn, transposed=True)

        def dfn(a, b, wr=wr, ww=ww, sh=sh):
            y = F.conv_transpose2d(a, wr, None, 2, 1) + F.conv_transpose2d(b, ww, None, 2, 1)
            return _rb(F.relu(y + sh.view(1, -1, 1, 1)))
        ops.append(('deconv.%d' % i, [refined, raw], dfn))
        refined = 'deconv.%d' % i
        raw = x_list[-i - 3]
        if i > 0:
            pr, pw = 'final_refined.%d.conv' % (i - 1), 'final_raw.%d.conv' % (i - 1)
            ca = d['heads'][i - 1]['refined_in']
            cb = d['heads'][i - 1]['raw_in']
            a = conv('final_refined.%d.dw5' % (i - 1), refined, pr + '.0.weight', pr + '.1', 1, 2, ca, 'relu')
            bq = conv('final_raw.%d.dw5' % (i - 1), raw, pw + '.0.weight', pw + '.1', 1, 2, cb, 'relu')
            w3a, _ = _fold(sd, pr + '.3.weight', None)
            w3b, _ = _fold(sd, pw + '.3.weight', None)

            def hfn(a, b, w3a=w3a, w3b=w3b):                # fp32 out: the AE stage reads fp32 maps
                return F.conv2d(a, w3a) + F.conv2d(b, w3b)
            ops.append(('final.%d.pw' % (i - 1), [a, bq], hfn))
    return ops


BF16_TAP_ALIASES = {'stem.pw': 'first'}


def forward_bf16(x, sd, arch, head=None, taps=None):
    """bf16-storage emulation of ``forward`` (same return value; ``taps`` receives every op output under the
    device op name plus the block-boundary names of ``forward``: 'first', 'stage.S.B', 'deconv.I')."""
    vals = {'x': x}
    outs = []
    for name, ins, fn in bf16_plan(sd, arch, head):
        y = fn(*[vals[k] for k in ins])
        vals[name] = y
        if name.startswith('final.') and name.endswith('.pw'):
            outs.append(y)
    if taps is not None:
        for k, v in vals.items():
            if k == 'x':
                continue
            taps[k] = v
            if k == 'stem.pw':
                taps['first'] = v
            elif k.endswith('.point_conv'):
                taps[k[:-len('.point_conv')]] = v
    return outs

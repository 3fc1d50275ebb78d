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

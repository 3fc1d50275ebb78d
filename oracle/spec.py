"""Arch spec -> channel bookkeeping and reference ``state_dict`` key scheme.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Restates, independently of the
product package, what ``LitePose.__init__`` derives from the arch JSON:
  * /root/reference/lib/models/pose_mobilenet.py:12-19   _make_divisible
  * /root/reference/lib/models/pose_mobilenet.py:31-60   stem + stages
  * /root/reference/lib/models/pose_mobilenet.py:86-100  final (head) layers
  * /root/reference/lib/models/pose_mobilenet.py:102-135 deconv layers
  * /root/reference/lib/models/layers/layers.py:90-133   InvBottleneck / SepConv2d
"""
from collections import OrderedDict


def make_divisible(v, divisor=8, min_value=None):
    # pose_mobilenet.py:12-19
    if min_value is None:
        min_value = divisor
    new_v = max(min_value, int(v + divisor / 2) // divisor * divisor)
    if new_v < 0.9 * v:
        new_v += divisor
    return new_v


class HeadCfg(object):
    """The handful of yacs fields that shape the network (mobile.yaml)."""

    def __init__(self, num_joints=14, tag_per_joint=True,
                 with_heatmaps_loss=(True, True), with_ae_loss=(True, False),
                 num_deconv_layers=3, deconv_kernels=(4, 4, 4)):
        self.num_joints = num_joints
        self.tag_per_joint = tag_per_joint
        self.with_heatmaps_loss = tuple(with_heatmaps_loss)
        self.with_ae_loss = tuple(with_ae_loss)
        self.num_deconv_layers = num_deconv_layers
        self.deconv_kernels = tuple(deconv_kernels)


def derive(arch, head=None):
    """Return a dict describing every layer the reference module would build."""
    head = head or HeadCfg()
    c0 = make_divisible(arch['input_channel'] * 1.0, 8)
    channel = [c0]
    stages = []
    inp = c0
    for st in arch['backbone_setting']:
        c = make_divisible(st['channel'] * 1.0, 8)
        blocks = []
        for b in range(st['num_blocks']):
            t, k = st['block_setting'][b]
            stride = st['stride'] if b == 0 else 1
            feat = make_divisible(round(inp * t), 8)          # layers.py:94
            blocks.append(dict(inp=inp, feat=feat, oup=c, k=k, stride=stride,
                               residual=(stride == 1 and inp == c)))
            inp = c
        stages.append(blocks)
        channel.append(c)
    filters = list(arch['deconv_setting'])
    deconv = []
    inplanes = channel[-1]
    for i in range(head.num_deconv_layers):
        assert head.deconv_kernels[i] == 4, 'only k=4,s=2,p=1 deconvs are on the path'
        deconv.append(dict(refined_in=inplanes, raw_in=channel[-i - 2], out=filters[i]))
        inplanes = filters[i]
    heads = []
    dim_tag = head.num_joints if head.tag_per_joint else 1
    for i in range(1, head.num_deconv_layers):
        oup = (head.num_joints if head.with_heatmaps_loss[i - 1] else 0) + \
              (dim_tag if head.with_ae_loss[i - 1] else 0)
        heads.append(dict(refined_in=filters[i], raw_in=channel[-i - 3], oup=oup))
    return dict(c0=c0, channel=channel, stages=stages, deconv=deconv, heads=heads)


def _bn_keys(prefix, c, out):
    out[prefix + '.weight'] = (c,)
    out[prefix + '.bias'] = (c,)
    out[prefix + '.running_mean'] = (c,)
    out[prefix + '.running_var'] = (c,)
    out[prefix + '.num_batches_tracked'] = ()


def state_dict_shapes(arch, head=None):
    """OrderedDict key -> shape, in the reference module's registration order
    (verified against the real module by gen_golden.py)."""
    d = derive(arch, head)
    o = OrderedDict()
    o['first.0.0.weight'] = (32, 3, 3, 3)
    _bn_keys('first.0.1', 32, o)
    o['first.1.0.weight'] = (32, 1, 3, 3)
    _bn_keys('first.1.1', 32, o)
    o['first.2.weight'] = (d['c0'], 32, 1, 1)
    _bn_keys('first.3', d['c0'], o)
    for s, blocks in enumerate(d['stages']):
        for b, blk in enumerate(blocks):
            p = 'stage.%d.%d' % (s, b)
            o[p + '.inv.0.weight'] = (blk['feat'], blk['inp'], 1, 1)
            _bn_keys(p + '.inv.1', blk['feat'], o)
            o[p + '.depth_conv.0.weight'] = (blk['feat'], 1, blk['k'], blk['k'])
            _bn_keys(p + '.depth_conv.1', blk['feat'], o)
            o[p + '.point_conv.0.weight'] = (blk['oup'], blk['feat'], 1, 1)
            _bn_keys(p + '.point_conv.1', blk['oup'], o)
    for i, dc in enumerate(d['deconv']):
        o['deconv_refined.%d.weight' % i] = (dc['refined_in'], dc['out'], 4, 4)
    for i, dc in enumerate(d['deconv']):
        o['deconv_raw.%d.weight' % i] = (dc['raw_in'], dc['out'], 4, 4)
    for i, dc in enumerate(d['deconv']):
        _bn_keys('deconv_bnrelu.%d.0' % i, dc['out'], o)
    for name, key in (('final_refined', 'refined_in'), ('final_raw', 'raw_in')):
        for i, h in enumerate(d['heads']):
            p = '%s.%d.conv' % (name, i)
            o[p + '.0.weight'] = (h[key], 1, 5, 5)
            _bn_keys(p + '.1', h[key], o)
            o[p + '.3.weight'] = (h['oup'], h[key], 1, 1)
    return o

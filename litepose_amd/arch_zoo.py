"""Architecture specs of the published LitePose models, in the reference's arch-dict
schema (``mobile_configs/*.json``: img_size, input_channel, deconv_setting,
backbone_setting[{num_blocks, stride, channel, block_setting[[t, k], ...]}]).

The JSON files themselves are an input format: ``load_arch(path)`` reads any file
with that schema (what ``valid.py --superconfig`` takes, valid.py:66-69,103-111).
The table below regenerates the seven published specs so that tests and the
benchmark do not need the reference checkout.
"""
import json

_NUM_BLOCKS = (6, 8, 10, 10)
_STRIDES = (2, 2, 2, 1)
#            name        img  in_ch  stage channels        deconv filters
_TABLE = {
    'search-XS': (256, 16, (16, 32, 48, 80), (16, 24, 24)),
    'search-S': (448, 16, (16, 32, 48, 120), (32, 24, 32)),
    'search-M': (448, 16, (24, 48, 72, 120), (64, 40, 32)),
    'search-L': (512, 24, (24, 64, 96, 160), (64, 40, 32)),
    'prune-S': (512, 16, (16, 32, 48, 80), (32, 24, 16)),
    'prune-M': (512, 24, (24, 48, 72, 120), (48, 40, 24)),
    'prune-L': (512, 24, (32, 64, 96, 160), (64, 48, 32)),
}


def names():
    return sorted(_TABLE)


def get(name):
    img, cin, chans, deconv = _TABLE[name]
    return {
        'img_size': img,
        'input_channel': cin,
        'deconv_setting': list(deconv),
        'backbone_setting': [
            {'num_blocks': n, 'stride': s, 'channel': c, 'block_setting': [[6, 7] for _ in range(n)]}
            for n, s, c in zip(_NUM_BLOCKS, _STRIDES, chans)
        ],
    }


def load_arch(path_or_name):
    if path_or_name in _TABLE:
        return get(path_or_name)
    with open(path_or_name, 'r') as f:
        return json.load(f)

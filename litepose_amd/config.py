"""Attribute-style config with the reference's keys (yacs is not a dependency).

Defaults = lib/config/default.py:20-153 overlaid with
experiments/crowd_pose/mobilenet/mobile.yaml (or the COCO variant); only keys the
inference path reads are kept.  ``merge_from_file`` accepts the reference's YAML
files unchanged (unknown keys are stored, not rejected).
"""
import copy

import yaml


class CfgNode(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def clone(self):
        return _wrap(copy.deepcopy(_unwrap(self)))

    def merge_from_dict(self, d):
        for k, v in d.items():
            if isinstance(v, dict) and isinstance(self.get(k), dict):
                self[k].merge_from_dict(v)
            else:
                self[k] = _wrap(v)

    def merge_from_file(self, path):
        with open(path, 'r') as f:
            self.merge_from_dict(yaml.safe_load(f))

    def merge_from_list(self, opts):
        """``KEY.SUB VALUE`` pairs, as valid.py's trailing CLI opts (default.py:156-160)."""
        assert len(opts) % 2 == 0
        for k, v in zip(opts[0::2], opts[1::2]):
            node = self
            parts = k.split('.')
            for p in parts[:-1]:
                node = node[p]
            node[parts[-1]] = yaml.safe_load(v) if isinstance(v, str) else v

    # yacs API no-ops so reference-style drivers run unchanged
    def defrost(self):
        pass

    def freeze(self):
        pass


def _wrap(v):
    if isinstance(v, dict) and not isinstance(v, CfgNode):
        return CfgNode({k: _wrap(x) for k, x in v.items()})
    return v


def _unwrap(v):
    if isinstance(v, dict):
        return {k: _unwrap(x) for k, x in v.items()}
    return v


_DEFAULT = {
    'GPUS': (0,),
    'DIST_BACKEND': 'nccl',
    'FP16': {'ENABLED': False},
    'DATASET': {
        'DATASET': 'crowd_pose_kpt', 'NUM_JOINTS': 14, 'MAX_NUM_PEOPLE': 30,
        'INPUT_SIZE': 256, 'OUTPUT_SIZE': [64, 128], 'WITH_CENTER': False, 'SIGMA': 2,
    },
    'LOSS': {'NUM_STAGES': 2, 'WITH_HEATMAPS_LOSS': [True, True], 'WITH_AE_LOSS': [True, False]},
    'MODEL': {
        'NAME': 'pose_mobilenet', 'NUM_JOINTS': 14, 'TAG_PER_JOINT': True, 'INIT_WEIGHTS': False,
        'PRETRAINED': '',
        'EXTRA': {'NUM_DECONV_LAYERS': 3, 'NUM_DECONV_KERNELS': [4, 4, 4],
                  'NUM_DECONV_FILTERS': [64, 48, 32], 'FINAL_CONV_KERNEL': 1},
    },
    'TEST': {
        'FLIP_TEST': True, 'ADJUST': True, 'REFINE': True, 'SCALE_FACTOR': [1],
        'DETECTION_THRESHOLD': 0.1, 'TAG_THRESHOLD': 1.0, 'USE_DETECTION_VAL': True,
        'IGNORE_TOO_MUCH': False, 'IGNORE_CENTER': True, 'MODEL_FILE': '',
        'NMS_KERNEL': 5, 'NMS_PADDING': 2, 'PROJECT2IMAGE': True,
        'WITH_HEATMAPS': (True, True), 'WITH_AE': (True, False), 'LOG_PROGRESS': False,
        'IMAGES_PER_GPU': 1,
    },
}


def get_cfg(dataset='crowd_pose'):
    """dataset: 'crowd_pose' (J=14, mobile.yaml) or 'coco' (J=17)."""
    cfg = _wrap(copy.deepcopy(_DEFAULT))
    if dataset.startswith('coco'):
        cfg.DATASET.DATASET = 'coco_kpt'
        cfg.DATASET.NUM_JOINTS = 17
        cfg.MODEL.NUM_JOINTS = 17
    return cfg


def apply_arch(cfg, arch):
    """valid.py:103-111: the arch JSON's img_size overrides INPUT_SIZE / OUTPUT_SIZE."""
    reso = arch['img_size']
    cfg.DATASET.INPUT_SIZE = reso
    cfg.DATASET.OUTPUT_SIZE = [reso // 4, reso // 2]
    return cfg


def enable_center(cfg, ignore_center=True):
    """lib/config/default.py:94,136,173-177 (update_config): DATASET.WITH_CENTER adds a centre joint to
    every stage (NUM_JOINTS += 1, MODEL.NUM_JOINTS follows); TEST.IGNORE_CENTER drops it again after
    the flip merge and in the parser.  Without IGNORE_CENTER only COCO (18 joints) has a joint order."""
    if not cfg.DATASET.WITH_CENTER:
        cfg.DATASET.WITH_CENTER = True
        cfg.DATASET.NUM_JOINTS = int(cfg.DATASET.NUM_JOINTS) + 1
        cfg.MODEL.NUM_JOINTS = cfg.DATASET.NUM_JOINTS
    cfg.TEST.IGNORE_CENTER = bool(ignore_center)
    return cfg


def update_config(cfg, args_or_path, opts=None):
    """lib/config/default.py:156-195 for the keys the inference path reads: merge the experiment YAML
    (``args.cfg``) and the trailing ``KEY VALUE`` opts (``args.opts``), then the reference's
    post-processing -- DATASET.WITH_CENTER adds the centre joint (NUM_JOINTS += 1, MODEL.NUM_JOINTS
    follows, :173-175) and scalar OUTPUT_SIZE / WITH_HEATMAPS_LOSS / WITH_AE_LOSS become lists (:177-186).
    ``args_or_path``: an argparse-style object with ``.cfg`` / ``.opts`` (what valid.py passes) or a path.
    Path joins with DATA_DIR (:161-171) are dataset/checkpoint plumbing and stay with the caller."""
    path = getattr(args_or_path, 'cfg', args_or_path)
    if opts is None:
        opts = getattr(args_or_path, 'opts', None)
    cfg.defrost()
    if path:
        cfg.merge_from_file(path)
    if opts:
        cfg.merge_from_list(list(opts))
    if cfg.DATASET.WITH_CENTER:
        cfg.DATASET.NUM_JOINTS = int(cfg.DATASET.NUM_JOINTS) + 1
        cfg.MODEL.NUM_JOINTS = cfg.DATASET.NUM_JOINTS
    if not isinstance(cfg.DATASET.OUTPUT_SIZE, (list, tuple)):
        cfg.DATASET.OUTPUT_SIZE = [cfg.DATASET.OUTPUT_SIZE]
    for key in ('WITH_HEATMAPS_LOSS', 'WITH_AE_LOSS'):
        if not isinstance(cfg.LOSS[key], (list, tuple)):
            cfg.LOSS[key] = [cfg.LOSS[key]]
    cfg.freeze()
    return cfg
